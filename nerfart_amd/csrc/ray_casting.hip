// ray_casting.hip - the per-ray stages of the surface renderer (reference models/ray_casting.py): pure
// consumers of the SDF kernel K2 (nerfart_sdf_fwd_rays evaluates the network on the marching / refinement points).
//
//   k_first_crossing   root_finding_surface_points (ray_casting.py:79-126): one wave per ray over the N_steps marched values:
//                      the first sign change of (sdf - tau), whether it goes outside -> inside, whether the ray starts outside,
//                      the bracketing (depth, value) pairs and the first secant estimate.
//   k_secant_update    one iteration of run_secant_method (ray_casting.py:11-30) for every bracketed ray.
//   k_root_finish      assembles depth / point / fill values (ray_casting.py:137-152).
//   k_sphere_step      one iteration of sphere_tracing_surface_points (ray_casting.py:175-180).
#include "nerfart_common.h"
#include <cmath>

namespace nerfart {
namespace rc {

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

// val [R, N] raw sdf at depths d [R, N]; one wave per ray, lane l owns steps l, l + 64, ...
__global__ __launch_bounds__(256) void k_first_crossing(const float* __restrict__ val, const float* __restrict__ d, int R, int N, float tau,
                                                       unsigned char* __restrict__ mask, unsigned char* __restrict__ mask_sc,
                                                       unsigned char* __restrict__ mask0, float* __restrict__ brk /*[R,4]*/,
                                                       float* __restrict__ d_pred) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= R) return;
    const float* v = val + (size_t)r * N;
    // cost_i = sign(v_i v_{i+1}) (N - i), last column +1 (ray_casting.py:91-100): minimise (cost, index) lexicographically
    float best = INFINITY;
    int besti = 0x7fffffff;
    for (int i = lane; i < N; i += 64) {
        const float a = v[i] - tau;
        const float s = (i + 1 < N) ? sgn(a * (v[i + 1] - tau)) : 1.f;
        const float c = s * (float)(N - i);
        if (c < best) { best = c; besti = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if (lane == 0) {
        const int i2 = min(besti + 1, N - 1);
        const float f_high = v[besti] - tau, f_low = v[i2] - tau;
        const float d_high = d[(size_t)r * N + besti], d_low = d[(size_t)r * N + i2];
        const bool sc = best < 0.f, m0 = (v[0] - tau) > 0.f;
        const bool m = sc && (f_high > 0.f) && m0;
        mask[r] = m; mask_sc[r] = sc; mask0[r] = m0;
        brk[4 * r + 0] = d_low; brk[4 * r + 1] = f_low; brk[4 * r + 2] = d_high; brk[4 * r + 3] = f_high;
        d_pred[r] = m ? (-f_low * (d_high - d_low) / (f_high - f_low) + d_low) : 1.0f;
    }
}

__global__ __launch_bounds__(256) void k_secant_update(const float* __restrict__ f_mid_raw, int R, float tau, const unsigned char* __restrict__ mask,
                                                      float* __restrict__ brk, float* __restrict__ d_pred) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R || !mask[r]) return;
    const float f_mid = f_mid_raw[r] - tau, dp = d_pred[r];
    float d_low = brk[4 * r], f_low = brk[4 * r + 1], d_high = brk[4 * r + 2], f_high = brk[4 * r + 3];
    if (f_mid < 0.f) { d_low = dp; f_low = f_mid; } else { d_high = dp; f_high = f_mid; }
    brk[4 * r] = d_low; brk[4 * r + 1] = f_low; brk[4 * r + 2] = d_high; brk[4 * r + 3] = f_high;
    d_pred[r] = -f_low * (d_high - d_low) / (f_high - f_low) + d_low;
}

__global__ __launch_bounds__(256) void k_root_finish(const float* __restrict__ rays_o, const float* __restrict__ rays_dn, int R,
                                                    const unsigned char* __restrict__ mask, const unsigned char* __restrict__ mask0,
                                                    const float* __restrict__ d_pred, const float* __restrict__ far, float far_s, int fill_inf,
                                                    float* __restrict__ d_out, float* __restrict__ pt_out) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    float dv;
    if (mask[r]) {
        dv = d_pred[r];
#pragma unroll
        for (int c = 0; c < 3; ++c) pt_out[3 * r + c] = ray_point(rays_o[3 * r + c], rays_dn[3 * r + c], dv);
    } else {
        dv = fill_inf ? INFINITY : (far ? far[r] : far_s);
#pragma unroll
        for (int c = 0; c < 3; ++c) pt_out[3 * r + c] = 1.0f;
    }
    if (!mask0[r]) dv = 0.f;                        // the ray starts inside the surface
    d_out[r] = dv;
}

// d[mask] += sdf[mask]; mask[d > far] = False; mask[d < 0] = False
__global__ __launch_bounds__(256) void k_sphere_step(const float* __restrict__ sdf, int R, const float* __restrict__ far, float far_s,
                                                    float* __restrict__ d, unsigned char* __restrict__ mask) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    float dv = d[r];
    bool m = mask[r];
    if (m) dv += sdf[r];
    if (dv > (far ? far[r] : far_s) || dv < 0.f) m = false;
    d[r] = dv;
    mask[r] = m;
}

}  // namespace rc
}  // namespace nerfart

using namespace nerfart;

extern "C" {

int nerfart_first_crossing(const float* val, const float* depth, int n_rays, int n_steps, float logit_tau, unsigned char* mask,
                           unsigned char* mask_sign_change, unsigned char* mask_start_outside, float* bracket, float* d_pred, void* stream) {
    if (n_rays <= 0) return 0;
    if (n_steps < 2) { set_last_error("first_crossing: need n_steps >= 2"); return 1; }
    hipLaunchKernelGGL(rc::k_first_crossing, dim3((n_rays + 3) / 4), dim3(256), 0, (hipStream_t)stream, val, depth, n_rays, n_steps, logit_tau, mask,
                       mask_sign_change, mask_start_outside, bracket, d_pred);
    NERFART_HIP(hipGetLastError());
    return 0;
}
int nerfart_secant_update(const float* f_mid, int n_rays, float logit_tau, const unsigned char* mask, float* bracket, float* d_pred, void* stream) {
    if (n_rays <= 0) return 0;
    hipLaunchKernelGGL(rc::k_secant_update, dim3((n_rays + 255) / 256), dim3(256), 0, (hipStream_t)stream, f_mid, n_rays, logit_tau, mask, bracket, d_pred);
    NERFART_HIP(hipGetLastError());
    return 0;
}
int nerfart_root_finish(const float* rays_o, const float* rays_dn, int n_rays, const unsigned char* mask, const unsigned char* mask_start_outside,
                        const float* d_pred, const float* far, float far_s, int fill_inf, float* d_out, float* pt_out, void* stream) {
    if (n_rays <= 0) return 0;
    hipLaunchKernelGGL(rc::k_root_finish, dim3((n_rays + 255) / 256), dim3(256), 0, (hipStream_t)stream, rays_o, rays_dn, n_rays, mask,
                       mask_start_outside, d_pred, far, far_s, fill_inf, d_out, pt_out);
    NERFART_HIP(hipGetLastError());
    return 0;
}
int nerfart_sphere_trace_step(const float* sdf, int n_rays, const float* far, float far_s, float* d, unsigned char* mask, void* stream) {
    if (n_rays <= 0) return 0;
    hipLaunchKernelGGL(rc::k_sphere_step, dim3((n_rays + 255) / 256), dim3(256), 0, (hipStream_t)stream, sdf, n_rays, far, far_s, d, mask);
    NERFART_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
