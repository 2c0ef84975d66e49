// capi_common.cpp - error reporting shared by every entry point of libnerfart_hip.so.
#include "nerfart_common.h"
#include <string>
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

// ---- optional launch profiling: HIP events around every chained-MLP launch ---------------------
// Used by bench.py to measure, live and on the launching stream, the average duration of the
// dominant kernel (roofline.achieved = algorithmic flops per launch / that duration).
namespace nerfart {
struct ProfRec { int cls; long long units; hipEvent_t a, b; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
bool profile_enabled() { return g_prof_on; }
void profile_open(int cls, long long units, hipStream_t s, void** handle) {
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
// c = 3 (k_wgrad<256>): units = algorithmic bytes (both operands once).  Host arrays of 4.
int nerfart_profile_end(double* ms, long long* launches, long long* units) {
    nerfart::g_prof_on = false;
    for (int c = 0; c < 4; ++c) { ms[c] = 0.0; launches[c] = 0; units[c] = 0; }
    for (auto& r : nerfart::g_prof) {
        float t = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess && r.cls >= 0 && r.cls < 4) {
            ms[r.cls] += t; launches[r.cls] += 1; units[r.cls] += r.units;
        }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    nerfart::g_prof.clear();
    return 0;
}
const char* nerfart_last_error(void) { return nerfart::g_last_error.c_str(); }
int nerfart_abi_version(void) { return 4; }
}
