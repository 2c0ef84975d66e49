// ubench_power.hip - what the MI355X SUSTAINS on v_mfma_f32_16x16x32_bf16 under its package power limit.
//
// Every wave runs a stream of MFMAs on 16 independent accumulator tiles (no dependency stalls), optionally with the fragment
// traffic of the MLP kernels beside it (two ds_read_b128 per three MFMAs, as one item of k_sdf_only_bf16) and its epilogue VALU
// (about 2 VALU per MFMA).  256 workgroups x 8 waves (2 waves per SIMD), each launch ~50 ms, launched back to back for several
// seconds per mode while the host samples the socket power (tools/power_probe_ubench.sh); prints achieved TFLOP/s (dense bf16) per mode
// for the first and for the last second of the run - the difference is the clock the power governor settles on.
//     hipcc --offload-arch=gfx950 -O3 -o tools/_build/ubench_power tools/ubench_power.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

template <int MODE>
__global__ __launch_bounds__(512, 2) void k_power(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned lds[16384];          // 64 KiB of "weights"
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = 0x3c003c00u + (i & 7);
    __syncthreads();
    f32x4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    u32x4 ar[4];
    if constexpr (MODE >= 3) {
        // "real" operands: pseudo-random bf16 values in [-2, 2) that differ per lane, per register and per MFMA (four A fragments
        // in rotation, as the weight tiles of a k-step do) - the multiplier arrays toggle as they do on trained weights
        unsigned h = (threadIdx.x + 1) * 2654435761u + blockIdx.x * 40503u;
        auto rnd = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return (h & 0x807f807fu) | 0x3f803f80u; };
        for (int q = 0; q < 4; ++q) ar[q] = u32x4{rnd(), rnd(), rnd(), rnd()};
        b = u32x4{rnd(), rnd(), rnd(), rnd()};
        a = ar[0];
    }
    u32x4 fa = a, fb = a;
    float v0 = threadIdx.x * 1e-3f, v1 = 1.0f, v2 = 0.5f, v3 = 0.25f;
    const unsigned addr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE >= 3) if ((it & 63) == 0) {
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if constexpr (MODE >= 1 && MODE != 3) {
                // one item of the MLP kernels: its two fragment reads (consumed two items later there; here the data is ignored
                // but the LDS traffic and the register writes are real), then three MFMAs on one accumulator chain
                asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(fa), "=&v"(fb) : "v"(addr), "i"((t & 15) * 2048), "i"((t & 15) * 2048 + 1024));
            }
            if constexpr (MODE >= 3) {
                MFMA(acc[t], ar[t & 3], b);
                MFMA(acc[t], ar[(t + 1) & 3], b);
                MFMA(acc[t], ar[(t + 2) & 3], b);
            } else {
                MFMA(acc[t], a, b);
                MFMA(acc[t], a, b);
                MFMA(acc[t], a, b);
            }
            if constexpr (MODE == 5) {
                // the same epilogue without the multiply in front of the exponential (the softplus scale folded into the
                // operands): 6 instead of 7 VALU per three MFMAs
                v0 = __builtin_amdgcn_exp2f(-v0);
                v1 = fmaf(v1, 0.999f, v0);
                v2 = __builtin_amdgcn_logf(1.0f + v0);
                v3 = fmaf(v3, 0.5f, v2);
                v1 = fmaf(v1, v3, 0.25f);
                v0 = v0 + v3;
            }
            if constexpr (MODE == 2 || MODE == 4) {
                // ~2 VALU per MFMA in the epilogue's proportions (exp2 / log2 / fma / cvt)
                v0 = __builtin_amdgcn_exp2f(v0 * -0.5f);
                v1 = fmaf(v1, 0.999f, v0);
                v2 = __builtin_amdgcn_logf(1.0f + v0);
                v3 = fmaf(v3, 0.5f, v2);
                v1 = fmaf(v1, v3, 0.25f);
                v0 = v0 + v3;
            }
            if constexpr (MODE >= 1 && MODE != 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa), "+v"(fb));
        }
    }
    float s = v0 + v1 + v2 + v3 + __uint_as_float(fa[0] ^ fb[1]) * 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][3];
    if (s == 123.456f) out[0] = s;                                       // keep everything live
}

template <int MODE>
static void run(const char* name, double seconds, float* out) {
    const int iters = 6000;                                              // 16 tiles x 3 MFMAs x 6000 = 288,000 MFMAs per wave per launch
    const double flop_per_launch = 256.0 * 8 * 288000.0 * (2.0 * 16 * 16 * 32);
    hipLaunchKernelGGL(k_power<MODE>, dim3(256), dim3(512), 0, 0, out, 100);
    hipDeviceSynchronize();
    std::vector<double> tf;
    auto t0 = std::chrono::steady_clock::now();
    double elapsed = 0;
    while (elapsed < seconds) {
        auto a = std::chrono::steady_clock::now();
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_power<MODE>, dim3(256), dim3(512), 0, 0, out, iters);
        hipDeviceSynchronize();
        auto b = std::chrono::steady_clock::now();
        tf.push_back(4 * flop_per_launch / std::chrono::duration<double>(b - a).count() / 1e12);
        elapsed = std::chrono::duration<double>(b - t0).count();
    }
    const size_t n = tf.size(), q = n / 5 ? n / 5 : 1;
    double first = 0, last = 0;
    for (size_t i = 0; i < q; ++i) { first += tf[i]; last += tf[n - 1 - i]; }
    printf("{\"mode\": \"%s\", \"seconds\": %.1f, \"tflops_first_fifth\": %.1f, \"tflops_last_fifth\": %.1f, \"frac_of_2500_sustained\": %.4f}\n", name, elapsed,
           first / q, last / q, last / q / 2500.0);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 6.0;
    float* out;
    hipMalloc(&out, 4096);
    printf("{\"marker\": \"idle\"}\n"); fflush(stdout);
    run<0>("mfma_only", seconds, out);
    run<1>("mfma_plus_fragment_reads", seconds, out);
    run<2>("mfma_plus_fragment_reads_plus_valu", seconds, out);
    run<3>("mfma_only_random_operands", seconds, out);
    run<4>("mfma_random_operands_plus_fragment_reads_plus_valu", seconds, out);
    run<5>("mode_4_with_one_valu_less_per_item", seconds, out);
    return 0;
}
