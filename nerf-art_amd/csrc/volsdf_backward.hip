// volsdf_backward.hip - backward of the per-ray stage of the VolSDF renderer (row a19, boundary B1 "bwd"):
// given d loss / d rgb per ray, the cotangents of every sample's sdf and radiance, and of (alpha, beta).
//
//   forward (volsdf.py:34-53, :544-576):  sigma_i = alpha psi(s_i; beta),  x_i = relu(sigma_i delta_i),  p_i = e^{-x_i},
//       T_i = prod_{j<i} p_j,  tau_i = (1 - p_i + 1e-10) T_i,  rgb = sum_i tau_i c_i  (+ 1 - sum tau, white background)
//   backward:  g_c_i = tau_i g_rgb;  g_tau_i = g_rgb . c_i (- sum g_rgb, white background)
//       g_x_j = p_j T_j g_tau_j - sum_{i>j} (1 - p_i + 1e-10) T_i g_tau_i       (no division by p_j: it underflows to 0)
//       g_sigma_j = [x_j > 0] delta_j g_x_j;  g_s_j = g_sigma_j alpha dpsi/ds,  dpsi/ds = -e^{-|s|/beta} / (2 beta)
//       g_alpha += g_sigma_j psi_j;  g_beta += g_sigma_j alpha dpsi/dbeta,  dpsi/dbeta = e^{-|s|/beta} s / (2 beta^2)
// One wave per ray (P - 1 intervals, in-lane sequential + 64-lane scans), like the forward kernel.  The last
// sample of a ray carries no weight (volsdf.py:556-560): its cotangents are 0.  Only the rgb cotangent is an
// input: the reference's losses do not touch depth / mask / normals maps.
#include "ray_common.h"

namespace nerfart {

// sum over the lanes AFTER this one (lane 63 gets 0)
__device__ __forceinline__ float wave_excl_suffix_sum(float v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_down(v, o, 64);
        if (lane + o < 64) v += t;
    }
    const float e = __shfl_down(v, 1, 64);
    return lane == 63 ? 0.f : e;
}

__global__ void __launch_bounds__(64)
k_composite_volsdf_bwd(int P, const float* __restrict__ d_all, const float* __restrict__ sdf,
                       const float* __restrict__ radiance, float alpha, float beta, int white_bkgd,
                       const float* __restrict__ g_rgb, float* __restrict__ g_sdf, float* __restrict__ g_rad,
                       float* __restrict__ g_alpha_beta) {
    const int ray = blockIdx.x, lane = threadIdx.x;
    const int nint = P - 1;
    const int seg = (nint + 63) >> 6;
    const int k0 = lane * seg, k1 = (k0 + seg < nint) ? k0 + seg : nint;
    const float* dr = d_all + (size_t)ray * P;
    const float* sr = sdf + (size_t)ray * P;
    const float gr = g_rgb[3 * (size_t)ray], gg = g_rgb[3 * (size_t)ray + 1], gb = g_rgb[3 * (size_t)ray + 2];
    const float gbg = white_bkgd ? -(gr + gg + gb) : 0.f;
    // pass 1: transmittance at the start of this lane's segment, and the segment's sum of (1 - p + eps) T g_tau
    float lp = 1.f;
    for (int k = k0; k < k1; ++k) lp *= expf(-fmaxf(sdf_to_sigma(sr[k], alpha, beta) * (dr[k + 1] - dr[k]), 0.f));
    const float T0 = wave_excl_prod(lp);
    float T = T0, hsum = 0.f;
    for (int k = k0; k < k1; ++k) {
        const float p = expf(-fmaxf(sdf_to_sigma(sr[k], alpha, beta) * (dr[k + 1] - dr[k]), 0.f));
        const size_t q = (size_t)ray * P + k;
        const float gtau = gr * radiance[3 * q] + gg * radiance[3 * q + 1] + gb * radiance[3 * q + 2] + gbg;
        hsum += (1.f - p + 1e-10f) * T * gtau;
        T *= p;
    }
    // S = sum over the intervals AFTER this lane's segment
    float S = wave_excl_suffix_sum(hsum);
    // pass 2 (descending inside the segment): cotangents
    float ga = 0.f, gbeta = 0.f;
    // transmittance at the END of the segment, walked backwards by recomputing forward values per element
    float Ts[8], ps[8];                       // seg <= 8 (P <= 513)
    T = T0;
    for (int k = k0, i = 0; k < k1; ++k, ++i) {
        const float p = expf(-fmaxf(sdf_to_sigma(sr[k], alpha, beta) * (dr[k + 1] - dr[k]), 0.f));
        Ts[i] = T; ps[i] = p;
        T *= p;
    }
    for (int k = k1 - 1, i = k1 - 1 - k0; k >= k0; --k, --i) {
        const size_t q = (size_t)ray * P + k;
        const float s = sr[k], delta = dr[k + 1] - dr[k];
        const float e = 0.5f * expf(-fabsf(s) / beta);
        const float psi = (s >= 0.f) ? e : 1.f - e;
        const float sg = alpha * psi;
        const float p = ps[i], Tk = Ts[i];
        const float c0 = radiance[3 * q], c1 = radiance[3 * q + 1], c2 = radiance[3 * q + 2];
        const float gtau = gr * c0 + gg * c1 + gb * c2 + gbg;
        const float tau = (1.f - p + 1e-10f) * Tk;
        g_rad[3 * q] = tau * gr; g_rad[3 * q + 1] = tau * gg; g_rad[3 * q + 2] = tau * gb;
        const float gx = p * Tk * gtau - S;
        const float gsig = (sg * delta > 0.f) ? gx * delta : 0.f;
        g_sdf[q] = gsig * alpha * (-e / beta);
        ga += gsig * psi;
        gbeta += gsig * alpha * (e * s / (beta * beta));
        S += tau * gtau;
    }
    if (lane == 0) {
        const size_t q = (size_t)ray * P + P - 1;
        g_sdf[q] = 0.f;
        g_rad[3 * q] = 0.f; g_rad[3 * q + 1] = 0.f; g_rad[3 * q + 2] = 0.f;
    }
    ga = wave_sum(ga); gbeta = wave_sum(gbeta);
    if (lane == 0 && g_alpha_beta) { atomicAdd(g_alpha_beta, ga); atomicAdd(g_alpha_beta + 1, gbeta); }
}

}  // namespace nerfart

using namespace nerfart;

extern "C" {
// g_alpha_beta: 2 floats, ACCUMULATED into (zero them first); may be NULL.
int nerfart_volsdf_composite_bwd(int n_rays, int P, const float* d_all, const float* sdf, const float* radiance, float alpha,
                                 float beta, int white_bkgd, const float* g_rgb, float* g_sdf, float* g_rad,
                                 float* g_alpha_beta, void* stream) {
    if (n_rays <= 0) return 0;
    if (P < 2 || P > 513) { set_last_error("composite_bwd: 2 <= P <= 513"); return 2; }
    hipLaunchKernelGGL(k_composite_volsdf_bwd, dim3(n_rays), dim3(64), 0, (hipStream_t)stream, P, d_all, sdf, radiance, alpha,
                       beta, white_bkgd, g_rgb, g_sdf, g_rad, g_alpha_beta);
    NERFART_HIP(hipGetLastError());
    return 0;
}
}
