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
                       const float* __restrict__ g_rgb, const float* __restrict__ g_acc, float* __restrict__ g_sdf,
                       float* __restrict__ g_rad,
                       float* __restrict__ g_alpha_beta) {
    const int ray = blockIdx.x, lane = threadIdx.x;
    const int nint = P - 1;
    const int seg = (nint + 63) >> 6;
    const int k0 = lane * seg, k1 = (k0 + seg < nint) ? k0 + seg : nint;
    const float* dr = d_all + (size_t)ray * P;
    const float* sr = sdf + (size_t)ray * P;
    const float gr = g_rgb[3 * (size_t)ray], gg = g_rgb[3 * (size_t)ray + 1], gb = g_rgb[3 * (size_t)ray + 2];
    // cotangent of the opacity acc = sum of the weights: the white background (rgb += 1 - acc) plus the caller's (mask loss)
    const float gbg = (white_bkgd ? -(gr + gg + gb) : 0.f) + (g_acc ? g_acc[ray] : 0.f);
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

// ---- NeuS (neus.py:29-78, :373-395): cdf_k = sigmoid(s sdf_k), alpha_k = max((cdf_k - cdf_{k+1}) / (cdf_k + 1e-10), 0),
// w_k = alpha_k prod_{j<k} (1 - alpha_j + 1e-10), rgb = sum_k w_k c_k with c at the interval mid-points.
//   g_c_k = w_k g_rgb;  g_w_k = g_rgb . c_k;  g_alpha_k = T_k g_w_k - (sum_{i>k} g_w_i w_i) / (1 - alpha_k + 1e-10)
//   alpha_k > 0:  g_cdf_k += g_alpha_k (cdf_{k+1} + 1e-10) / (cdf_k + 1e-10)^2,  g_cdf_{k+1} -= g_alpha_k / (cdf_k + 1e-10)
//   g_sdf_k = g_cdf_k cdf_k (1 - cdf_k) s;  g_s += g_cdf_k cdf_k (1 - cdf_k) sdf_k
__global__ void __launch_bounds__(64)
k_composite_neus_bwd(int P, const float* __restrict__ sdf, const float* __restrict__ rad_mid, float s_inv, int white_bkgd,
                     const float* __restrict__ g_rgb, const float* __restrict__ g_acc, float* __restrict__ g_sdf, float* __restrict__ g_rad, float* __restrict__ g_s) {
    const int ray = blockIdx.x, lane = threadIdx.x;
    const int nint = P - 1, seg = (nint + 63) >> 6, k0 = lane * seg, k1 = (k0 + seg < nint) ? k0 + seg : nint;
    const float* sr = sdf + (size_t)ray * P;
    const float gr = g_rgb[3 * (size_t)ray], gg = g_rgb[3 * (size_t)ray + 1], gb = g_rgb[3 * (size_t)ray + 2];
    // cotangent of the opacity acc = sum of the weights: the white background (rgb += 1 - acc) plus the caller's (mask loss)
    const float gbg = (white_bkgd ? -(gr + gg + gb) : 0.f) + (g_acc ? g_acc[ray] : 0.f);
    float lp = 1.f;
    for (int k = k0; k < k1; ++k) {
        const float c0 = sigmoidf_(sr[k] * s_inv), c1 = sigmoidf_(sr[k + 1] * s_inv);
        lp *= (1.f - fmaxf((c0 - c1) / (c0 + 1e-10f), 0.f) + 1e-10f);
    }
    const float T0 = wave_excl_prod(lp);
    float T = T0, hsum = 0.f;
    float Ts[8], as[8], c0s[8], c1s[8];                       // seg <= 8 (P <= 513)
    for (int k = k0, i = 0; k < k1; ++k, ++i) {
        const float c0 = sigmoidf_(sr[k] * s_inv), c1 = sigmoidf_(sr[k + 1] * s_inv);
        const float a = fmaxf((c0 - c1) / (c0 + 1e-10f), 0.f);
        const size_t q = (size_t)ray * nint + k;
        const float gw = gr * rad_mid[3 * q] + gg * rad_mid[3 * q + 1] + gb * rad_mid[3 * q + 2] + gbg;
        Ts[i] = T; as[i] = a; c0s[i] = c0; c1s[i] = c1;
        hsum += gw * a * T;
        T *= (1.f - a + 1e-10f);
    }
    float S = wave_excl_suffix_sum(hsum);
    float gc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) gc[i] = 0.f;
    for (int k = k1 - 1, i = k1 - 1 - k0; k >= k0; --k, --i) {
        const size_t q = (size_t)ray * nint + k;
        const float a = as[i], Tk = Ts[i], c0 = c0s[i], c1 = c1s[i];
        const float col0 = rad_mid[3 * q], col1 = rad_mid[3 * q + 1], col2 = rad_mid[3 * q + 2];
        const float gw = gr * col0 + gg * col1 + gb * col2 + gbg;
        const float wk = a * Tk;
        g_rad[3 * q] = wk * gr; g_rad[3 * q + 1] = wk * gg; g_rad[3 * q + 2] = wk * gb;
        const float ga = Tk * gw - S / (1.f - a + 1e-10f);
        if (a > 0.f) {
            const float inv = 1.f / (c0 + 1e-10f);
            gc[i] += ga * (c1 + 1e-10f) * inv * inv;
            gc[i + 1] -= ga * inv;
        }
        S += gw * wk;
    }
    // the cdf_{k+1} part of a segment's last interval belongs to the first sample of the next lane
    const int cnt = k1 - k0;
    float carry = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) if (i == cnt) carry = gc[i];
    float from_prev = __shfl_up(carry, 1, 64);
    if (lane == 0) from_prev = 0.f;
    // lanes past the last interval (cnt <= 0) only pass nothing on; sample P-1 is written by the lane that owns interval P-2
    float gs_acc = 0.f;
    for (int k = k0, i = 0; k < k1; ++k, ++i) {
        const float gck = gc[i] + (i == 0 ? from_prev : 0.f);
        const float c = c0s[i];
        const float t = gck * c * (1.f - c);
        g_sdf[(size_t)ray * P + k] = t * s_inv;
        gs_acc += t * sr[k];
    }
    if (cnt > 0 && k1 == nint) {
        const float c = c1s[cnt - 1];
        const float t = carry * c * (1.f - c);
        g_sdf[(size_t)ray * P + P - 1] = t * s_inv;
        gs_acc += t * sr[P - 1];
    }
    gs_acc = wave_sum(gs_acc);
    if (lane == 0 && g_s) atomicAdd(g_s, gs_acc);
}

}  // namespace nerfart

using namespace nerfart;

extern "C" {
// g_alpha_beta: 2 floats, ACCUMULATED into (zero them first); may be NULL.
int nerfart_volsdf_composite_bwd(int n_rays, int P, const float* d_all, const float* sdf, const float* radiance, float alpha,
                                 float beta, int white_bkgd, const float* g_rgb, const float* g_acc, float* g_sdf, float* g_rad,
                                 float* g_alpha_beta, void* stream) {
    if (n_rays <= 0) return 0;
    if (P < 2 || P > 513) { set_last_error("composite_bwd: 2 <= P <= 513"); return 2; }
    hipLaunchKernelGGL(k_composite_volsdf_bwd, dim3(n_rays), dim3(64), 0, (hipStream_t)stream, P, d_all, sdf, radiance, alpha,
                       beta, white_bkgd, g_rgb, g_acc, g_sdf, g_rad, g_alpha_beta);
    NERFART_HIP(hipGetLastError());
    return 0;
}

// NeuS: sdf [R,P], rad_mid [R,P-1,3], s = exp(ln_s * speed_factor); g_s: 1 float, ACCUMULATED into (may be NULL).
int nerfart_neus_composite_bwd(int n_rays, int P, const float* sdf, const float* rad_mid, float s, int white_bkgd, const float* g_rgb,
                               const float* g_acc, float* g_sdf, float* g_rad_mid, float* g_s, void* stream) {
    if (n_rays <= 0) return 0;
    if (P < 2 || P > 513) { set_last_error("neus_composite_bwd: 2 <= P <= 513"); return 2; }
    hipLaunchKernelGGL(k_composite_neus_bwd, dim3(n_rays), dim3(64), 0, (hipStream_t)stream, P, sdf, rad_mid, s, white_bkgd, g_rgb, g_acc,
                       g_sdf, g_rad_mid, g_s);
    NERFART_HIP(hipGetLastError());
    return 0;
}
}
