// capi_common.cpp - error reporting shared by every entry point of libnerfart_hip.so.
#include "nerfart_common.h"
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace nerfart {
static thread_local std::string g_last_error;
void set_last_error(const char* s) { g_last_error = s ? s : ""; }
int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return 1;
}
}  // namespace nerfart

// ---- which fragment encoding a blob carries: the packers record (device pointer -> header word 10: 0 fp32, 1 bf16 hi + lo, 2 fp16
// hi + lo) on the HOST, and the entry points that read ONE encoding look the pointer up before launching - a bf16 and an fp16 blob have the same
// size and program id, and reading header word 10 back from the device would cost a stream synchronisation per call.  A pointer the packers never
// saw (a caller's copy of a blob) is not checked; a pointer re-packed in place is re-recorded; a record that contradicts the call is verified
// against the blob's header on the device before the call is refused (a recycled address must not fail a valid blob).
namespace nerfart {
static std::mutex g_term_mu;
static std::unordered_map<const void*, int> g_term;
void blob_term_register(const void* blob, int term) {
    std::lock_guard<std::mutex> lk(g_term_mu);
    if (g_term.size() > 4096) g_term.clear();            // blobs are re-packed after every optimiser step, mostly into recycled addresses
    g_term[blob] = term;
}
int blob_term_check(const void* blob, int want, const char* who) {
    int have = -1;
    {
        std::lock_guard<std::mutex> lk(g_term_mu);
        auto it = g_term.find(blob);
        if (it != g_term.end()) have = it->second;
    }
    if (have < 0 || have == want) return 0;
    // The record may be STALE (the address was a blob once, was freed, and now holds the caller's copy of another blob): before refusing, read the
    // blob's own header word 10 - a device synchronisation, paid only on this suspected-mismatch path - and believe the header.
    int hdr10 = -1;
    if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(&hdr10, (const int*)blob + 10, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess &&
        hdr10 >= 0 && hdr10 <= 3) {
        blob_term_register(blob, hdr10);
        if (hdr10 == want) return 0;
        have = hdr10;
    }
    static const char* names[] = {"fp32", "split bf16 (precision 1)", "fp16 hi + lo (precision 4)", "softplus-scaled fp16 (precision 5: the 1-MFMA sampler's)"};
    g_last_error = std::string(who) + ": this entry point reads a " + names[want < 4 ? want : 0] + " blob; the blob was packed as " + names[have < 4 ? have : 0] +
                   " (nerfart_pack_*_blob's precision argument)";
    return 2;
}
}  // namespace nerfart

// ---- optional launch profiling: HIP events around every chained-MLP launch ---------------------
// Used by bench.py to measure, live and on the launching stream, the average duration of the
// dominant kernel (roofline.achieved = algorithmic flops per launch / that duration).
namespace nerfart {
struct ProfRec { int cls; long long units; hipEvent_t a, b; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static thread_local int g_cls0_as = 0;          // fine_sample_run's escalation run: its SDF queries (class 0) are recorded as class 4
void profile_class0_as(int cls) { g_cls0_as = cls; }
bool profile_enabled() { return g_prof_on; }
void profile_open(int cls, long long units, hipStream_t s, void** handle) {
    if (cls == 0 && g_cls0_as) cls = g_cls0_as;
    ProfRec r{cls, units, nullptr, nullptr};
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) { *handle = nullptr; return; }
    (void)hipEventRecord(r.a, s);
    g_prof.push_back(r);
    *handle = (void*)(size_t)g_prof.size();
}
void profile_close(void* handle, hipStream_t s) {
    if (!handle) return;
    (void)hipEventRecord(g_prof[(size_t)handle - 1].b, s);
}
}  // namespace nerfart

extern "C" {
int nerfart_profile_begin(void) {
    for (auto& r : nerfart::g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    nerfart::g_prof.clear();
    nerfart::g_prof_on = true;
    return 0;
}
// ms[c], launches[c], units[c] for c = 0 (k_sdf_only), 1 (k_sdf_nabla / k_sdf_grad), 2 (k_radiance): units = points;
// c = 3 (k_wgrad<256>): units = algorithmic bytes (both operands once); c = 4 (ABI 4, nerfart_profile_end5 only): the SDF queries of the guarded
// sampler's ESCALATION run (the same kernels as class 0 at the escalation precision, on the re-sampled rays), kept apart so that class 0 stays the
// dominant kernel's own launches.  Host arrays of 4 (nerfart_profile_end: class 4 is dropped) / 5.
static int profile_end_n(double* ms, long long* launches, long long* units, int n) {
    nerfart::g_prof_on = false;
    for (int c = 0; c < n; ++c) { ms[c] = 0.0; launches[c] = 0; units[c] = 0; }
    for (auto& r : nerfart::g_prof) {
        float t = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess && r.cls >= 0 && r.cls < n) {
            ms[r.cls] += t; launches[r.cls] += 1; units[r.cls] += r.units;
        }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    nerfart::g_prof.clear();
    return 0;
}
int nerfart_profile_end(double* ms, long long* launches, long long* units) { return profile_end_n(ms, launches, units, 4); }
int nerfart_profile_end5(double* ms, long long* launches, long long* units) { return profile_end_n(ms, launches, units, 5); }
const char* nerfart_last_error(void) { return nerfart::g_last_error.c_str(); }
int nerfart_abi_version(void) { return 5; }
}
