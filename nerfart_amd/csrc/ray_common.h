// ray_common.h - wave-per-ray building blocks shared by the VolSDF / NeuS sampler and compositor
// kernels.  One 64-lane wave owns one ray; the ray's samples live in LDS; prefix sums are
// "sequential inside a lane's contiguous segment + shuffle scan across the 64 lanes".
// (Round 2 experiment, reverted: odd segment lengths - seg = ceil(n / 64) | 1 - make lane l's addresses l * seg + i hit 32
// different banks: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE fell from 86 / 73 / 52 % to 8 / 0 / 2 % in k_merge_check / k_first_check /
// k_upsample (profiles/r02n_pmc_lds.txt) but the kernels' times did not move (1.27 -> 1.22 ms, 0.55 -> 0.54 ms: they wait on
// dependent expf chains and binary searches, not on LDS bandwidth), and the regrouped prefix sums flipped a bisection branch of one
// golden ray past the 1e-3 pixel bound in the split-bf16 mode.  A conflict-free layout has to keep the partition and pad addresses.
// Round 6: the conflicts were taken out of the loop that runs 11 times per ray instead - SegConsts below - with the same partition and the same bits:
// k_merge_check 1.13 -> 1.07 ms per launch; the kernel waits on its exp / divide chains, as round 2 found.)
#pragma once
#include "nerfart_common.h"
#include <math.h>

namespace nerfart {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// exclusive prefix sum over lanes (lane 0 gets 0)
__device__ __forceinline__ float wave_excl_sum(float v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    const float e = __shfl_up(v, 1, 64);
    return lane == 0 ? 0.f : e;
}
// exclusive prefix product over lanes (lane 0 gets 1)
__device__ __forceinline__ float wave_excl_prod(float v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(v, o, 64);
        if (lane >= o) v *= t;
    }
    const float e = __shfl_up(v, 1, 64);
    return lane == 0 ? 1.f : e;
}

// a11 sdf_to_sigma (reference models/frameworks/volsdf.py:34-53)
__device__ __forceinline__ float sdf_to_sigma(float s, float alpha, float beta) {
    const float e = 0.5f * expf(-fabsf(s) / beta);
    return alpha * ((s >= 0.f) ? e : 1.f - e);
}

// first index i in [0, n] with c[i] >= u  (torch.searchsorted(..., right=False))
__device__ __forceinline__ int lower_bound(const float* c, int n, float u) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (c[mid] < u) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// first index i in [0, n] with c[i] > u
__device__ __forceinline__ int upper_bound(const float* c, int n, float u) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (c[mid] <= u) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Piece-wise linear inverse CDF of one sample (utils/rend_util.py:276-291): bracket by lower
// bound, clamp to the array, a denominator below 1e-5 becomes 1.
__device__ __forceinline__ float invert_cdf_at(const float* bins, const float* cdf, int n, float u) {
    const int idx = lower_bound(cdf, n, u);
    const int below = idx - 1 < 0 ? 0 : idx - 1;
    const int above = idx > n - 1 ? n - 1 : idx;
    const float c0 = cdf[below], c1 = cdf[above];
    float denom = c1 - c0;
    if (denom < 1e-5f) denom = 1.f;
    const float t = (u - c0) / denom;
    const float b0 = bins[below], b1 = bins[above];
    return b0 + t * (b1 - b0);
}

// a12 error_bound (volsdf.py:56-94) of the n-1 intervals of one ray held in LDS.
// Returns the maximum bound (wave-uniform).  If w_out != nullptr, bound k is also written to
// w_out[k] (after NaN -> +inf and, if clamp, clamp to [0, 1e5] as volsdf.py:282).
// Two passes over a lane's contiguous segment of intervals: local sums of sigma delta and of the E terms, a shuffle scan across the lanes, then
// the bounds with the running sums.  GENERIC form: any segment length, the second pass recomputes the first pass's two exponentials.
__device__ __forceinline__ float error_bound_scan_generic(const float* d, const float* s, int n, float alpha, float beta,
                                                          float* w_out, bool clamp) {
    const int lane = threadIdx.x & 63;
    const int nint = n - 1;
    const int seg = (nint + 63) >> 6;
    const int k0 = lane * seg;
    const int k1 = (k0 + seg < nint) ? k0 + seg : nint;
    const float a4b = alpha / (4.f * beta);
    float sR = 0.f, sE = 0.f;
    for (int k = k0; k < k1; ++k) {
        const float delta = d[k + 1] - d[k];
        sR += sdf_to_sigma(s[k], alpha, beta) * delta;
        const float dstar = fmaxf(0.5f * (fabsf(s[k]) + fabsf(s[k + 1]) - delta), 0.f);
        sE += a4b * (delta * delta) * expf(-dstar / beta);
    }
    float R = wave_excl_sum(sR), E = wave_excl_sum(sE);
    float mx = -INFINITY;
    for (int k = k0; k < k1; ++k) {
        const float delta = d[k + 1] - d[k];
        const float sd = sdf_to_sigma(s[k], alpha, beta) * delta;
        const float dstar = fmaxf(0.5f * (fabsf(s[k]) + fabsf(s[k + 1]) - delta), 0.f);
        E += a4b * (delta * delta) * expf(-dstar / beta);
        float b = expf(-R) * (expf(E) - 1.f);
        if (isnan(b)) b = INFINITY;
        if (clamp) b = fminf(fmaxf(b, 0.f), 1e5f);
        if (w_out) w_out[k] = b;
        mx = fmaxf(mx, b);
        R += sd;
    }
    return wave_max(mx);
}

// The same for segments of at most MAXSEG intervals (round 6): the first pass keeps each interval's four factors - delta, sigma,
// alpha / (4 beta) delta^2, exp(-d* / beta) - in registers, so the second pass neither re-reads the LDS rows (lane l's addresses l * seg + i are a
// seg-word stride: up to 32 lanes per bank) nor recomputes the two exponentials and two divisions per interval, and the fully unrolled first pass
// runs its MAXSEG independent exp / divide chains interleaved instead of one after the other.  BIT-IDENTICAL to the generic form: the
// accumulations are written as the fused multiply-adds hipcc contracts the generic form's `+=` into (checked in the ISA: v_fmac_f32 with exactly
// these operands in both passes), in the same order; tests/test_gpu_guarded_sampler.py::test_cached_scan_equals_generic_scan holds the two equal
// on whole frames (NERFART_SCAN_GENERIC builds the generic form everywhere).
template <int MAXSEG>
__device__ __forceinline__ float error_bound_scan_cached(const float* d, const float* s, int n, float alpha, float beta,
                                                         float* w_out, bool clamp) {
    const int lane = threadIdx.x & 63;
    const int nint = n - 1;
    const int seg = (nint + 63) >> 6;
    const int k0 = lane * seg;
    const int k1 = (k0 + seg < nint) ? k0 + seg : nint;
    const float a4b = alpha / (4.f * beta);
    float c_delta[MAXSEG], c_sigma[MAXSEG], c_t[MAXSEG], c_x[MAXSEG];
    float sR = 0.f, sE = 0.f;
#pragma unroll
    for (int i = 0; i < MAXSEG; ++i) {
        const int k = k0 + i;
        c_delta[i] = 0.f; c_sigma[i] = 0.f; c_t[i] = 0.f; c_x[i] = 0.f;
        if (k < k1) {
            const float delta = d[k + 1] - d[k];
            const float sg = sdf_to_sigma(s[k], alpha, beta);
            const float dstar = fmaxf(0.5f * (fabsf(s[k]) + fabsf(s[k + 1]) - delta), 0.f);
            const float t = a4b * (delta * delta);
            const float x = expf(-dstar / beta);
            c_delta[i] = delta; c_sigma[i] = sg; c_t[i] = t; c_x[i] = x;
            sR = __fmaf_rn(delta, sg, sR);
            sE = __fmaf_rn(t, x, sE);
        }
    }
    float R = wave_excl_sum(sR), E = wave_excl_sum(sE);
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXSEG; ++i) {
        const int k = k0 + i;
        if (k < k1) {
            E = __fmaf_rn(c_t[i], c_x[i], E);
            float b = expf(-R) * (expf(E) - 1.f);
            if (isnan(b)) b = INFINITY;
            if (clamp) b = fminf(fmaxf(b, 0.f), 1e5f);
            if (w_out) w_out[k] = b;
            mx = fmaxf(mx, b);
            R = __fmaf_rn(c_delta[i], c_sigma[i], R);
        }
    }
    return wave_max(mx);
}

// The beta-INDEPENDENT part of a lane's intervals, held in registers across scans (round 6 b): k_merge_check scans the same merged
// rows up to 11 times (the check at the net's beta + 10 bisection steps, volsdf.py:240-275) and only (alpha, beta) change in between - delta, s_k,
// d*_k and delta^2 do not.  Loaded once per ray (the one stride-seg pass over the LDS rows that is left: lane l's addresses l * seg + i put up to 32
// lanes on a bank), after which a scan touches no LDS at all.  scan_consts is error_bound_scan_cached with its first pass's loads and
// beta-independent arithmetic hoisted: the same operations on the same values in the same order (delta * delta and 0.5 (|s_k| + |s_k+1| - delta)
// are not contractable into their consumers), i.e. the same bits - tests/test_gpu_guarded_sampler.py::test_cached_scan_equals_generic_scan holds
// whole frames equal to the generic form.
template <int MAXSEG>
struct SegConsts { float delta[MAXSEG], s[MAXSEG], dstar[MAXSEG], dd[MAXSEG]; int cnt; };

template <int MAXSEG>
__device__ __forceinline__ void load_seg_consts(const float* d, const float* s, int n, SegConsts<MAXSEG>& C) {
    const int lane = threadIdx.x & 63;
    const int nint = n - 1;
    const int seg = (nint + 63) >> 6;
    const int k0 = lane * seg;
    const int k1 = (k0 + seg < nint) ? k0 + seg : nint;
    C.cnt = k1 - k0;
    float dk = 0.f, sk = 0.f;
    if (k0 < k1) { dk = d[k0]; sk = s[k0]; }
#pragma unroll
    for (int i = 0; i < MAXSEG; ++i) {
        C.delta[i] = 0.f; C.s[i] = 0.f; C.dstar[i] = 0.f; C.dd[i] = 0.f;
        if (i < C.cnt) {
            const float dn = d[k0 + i + 1], sn = s[k0 + i + 1];
            const float delta = dn - dk;
            C.delta[i] = delta;
            C.s[i] = sk;
            C.dstar[i] = fmaxf(0.5f * (fabsf(sk) + fabsf(sn) - delta), 0.f);
            C.dd[i] = delta * delta;
            dk = dn; sk = sn;
        }
    }
}

template <int MAXSEG>
__device__ __forceinline__ float scan_consts(const SegConsts<MAXSEG>& C, float alpha, float beta) {
    const float a4b = alpha / (4.f * beta);
    float c_sigma[MAXSEG], c_t[MAXSEG], c_x[MAXSEG];
    float sR = 0.f, sE = 0.f;
#pragma unroll
    for (int i = 0; i < MAXSEG; ++i) {
        c_sigma[i] = 0.f; c_t[i] = 0.f; c_x[i] = 0.f;
        if (i < C.cnt) {
            const float sg = sdf_to_sigma(C.s[i], alpha, beta);
            const float t = a4b * C.dd[i];
            const float x = expf(-C.dstar[i] / beta);
            c_sigma[i] = sg; c_t[i] = t; c_x[i] = x;
            sR = __fmaf_rn(C.delta[i], sg, sR);
            sE = __fmaf_rn(t, x, sE);
        }
    }
    float R = wave_excl_sum(sR), E = wave_excl_sum(sE);
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXSEG; ++i) {
        if (i < C.cnt) {
            E = __fmaf_rn(c_t[i], c_x[i], E);
            float b = expf(-R) * (expf(E) - 1.f);
            if (isnan(b)) b = INFINITY;
            mx = fmaxf(mx, b);
            R = __fmaf_rn(C.delta[i], c_sigma[i], R);
        }
    }
    return wave_max(mx);
}

__device__ __forceinline__ float error_bound_scan(const float* d, const float* s, int n, float alpha, float beta,
                                                  float* w_out, bool clamp) {
#ifndef NERFART_SCAN_GENERIC
    const int seg = (n - 1 + 63) >> 6;                       // wave-uniform: n is
    if (seg <= 8) return error_bound_scan_cached<8>(d, s, n, alpha, beta, w_out, clamp);          // 512 samples: the first check
    if (seg <= 16) return error_bound_scan_cached<16>(d, s, n, alpha, beta, w_out, clamp);        // 1,024: round 1 (every undecided ray of a frame)
    if (seg <= 24) return error_bound_scan_cached<24>(d, s, n, alpha, beta, w_out, clamp);        // 1,536: round 2 (78 % of them)
#endif
    return error_bound_scan_generic(d, s, n, alpha, beta, w_out, clamp);
}

// cdf[0] = 0, cdf[k+1] = 1 - exp(-R_t[k]) with R_t[k] = sum_{i<k} sigma_i delta_i, k = 0..n-2
// (opacity_invert_cdf_sample, volsdf.py:122-136 + the leading zero of sample_cdf, rend_util.py:298-300)
__device__ __forceinline__ void opacity_cdf(const float* d, const float* s, int n, float alpha, float beta, float* cdf) {
    const int lane = threadIdx.x & 63;
    const int nint = n - 1;
    const int seg = (nint + 63) >> 6;
    const int k0 = lane * seg;
    const int k1 = (k0 + seg < nint) ? k0 + seg : nint;
    float sR = 0.f;
    for (int k = k0; k < k1; ++k) sR += sdf_to_sigma(s[k], alpha, beta) * (d[k + 1] - d[k]);
    float R = wave_excl_sum(sR);
    if (lane == 0) cdf[0] = 0.f;
    for (int k = k0; k < k1; ++k) {
        cdf[k + 1] = 1.f - expf(-R);
        R += sdf_to_sigma(s[k], alpha, beta) * (d[k + 1] - d[k]);
    }
}

// In-LDS bitonic sort (ascending) of n = power-of-two floats by one wave.
__device__ __forceinline__ void bitonic_sort(float* a, int n) {
    const int lane = threadIdx.x & 63;
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int i = lane; i < n; i += 64) {
                const int p = i ^ j;
                if (p > i) {
                    const float x = a[i], y = a[p];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { a[i] = y; a[p] = x; }
                }
            }
        }
    }
    __syncthreads();
}

}  // namespace nerfart
