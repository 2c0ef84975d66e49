// volsdf_render.hip - VolSDF Algorithm-1 sampler, sample merge, compositor and the render
// orchestrator (reference models/frameworks/volsdf.py: error_bound :56-94, fine_sample :97-302,
// volume_render :389-615; utils/rend_util.py sample_pdf/sample_cdf :256-328).
//
// All per-ray kernels: one 64-lane wave (= one workgroup) per ray, the ray's samples in LDS.
// The data-dependent up-sampling loop keeps a compacted list of still-active rays; per round it
// runs   upsample (inverse CDF of the error bound, 512 new depths)
//     -> k_sdf_only over (active rays x 512 new points)           [mlp_chain.hip]
//     -> merge + bound check at the net's beta + (converged: 64 inverse-CDF samples of the opacity
//        | else: 10-step bisection for beta+ and re-queue).
// One 4-byte device->host read of the active count per round is the only host synchronisation.
#include "ray_common.h"
#include <stdio.h>
#include <stdlib.h>

namespace nerfart {

struct SamplerParams {
    int n;            // samples currently held per active ray
    int cap;          // row stride of the d/s arrays
    int n_up;         // new samples per round (512)
    int n_final;      // N_importance (64)
    int max_bisect;   // 10
    int it;           // round number written to iter_usage on convergence
    float eps;
    float alpha_net, beta_net;
    int u_final_stride;   // 0: one shared table u_final[n_final] (det = True: linspace); n_final: a row per ray (perturb: rand)
    // guard band of the convergence decision `max B > eps` (volsdf.py:162-163, :240-242): a ray whose max B lies within guard * eps of eps is
    // neither sampled nor re-queued here - it is appended to esc_list and Algorithm 1 runs again for it on the escalation blob (fine_sample_run)
    float guard;          // <= 0: off
    int* esc_list;
    int* esc_count;
};

// wave-uniform: is the decision `mx > eps` inside the guard band?  (mx = +inf - a NaN bound - is a clear "not converged")
__device__ __forceinline__ bool in_guard_band(const SamplerParams& P, float mx) {
    return P.guard > 0.f && fabsf(mx - P.eps) <= P.guard * P.eps;
}
__device__ __forceinline__ void escalate(const SamplerParams& P, int ray) {
    if (threadIdx.x == 0) P.esc_list[atomicAdd(P.esc_count, 1)] = ray;
}

// d = near * (1 - t) + far * t with the reference's three roundings (volsdf.py:474, :484)
__global__ void k_linspace_depths(const float* __restrict__ t, int n, const float* __restrict__ near,
                                  const float* __restrict__ far, float near_s, float far_s, int n_rays,
                                  float* __restrict__ out, int stride) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n_rays * n) return;
    const int r = (int)(i / n), k = (int)(i - (long long)r * n);
    const float nr = near ? near[r] : near_s, fr = far ? far[r] : far_s;
    const float tk = t[k];
    out[(size_t)r * stride + k] = __fadd_rn(__fmul_rn(nr, __fsub_rn(1.f, tk)), __fmul_rn(fr, tk));
}

// F.normalize(rays_d, dim=-1): v / max(||v||, 1e-12)   (volsdf.py:442)
__global__ void k_normalize_dirs(const float* __restrict__ in, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
    const float nrm = fmaxf(sqrtf(x * x + y * y + z * z), 1e-12f);
    out[3 * i] = x / nrm; out[3 * i + 1] = y / nrm; out[3 * i + 2] = z / nrm;
}

__device__ __forceinline__ void load_row(float* dst, const float* src, int n) {
    for (int i = threadIdx.x; i < n; i += 64) dst[i] = src[i];
}

__device__ __forceinline__ void emit_final_samples(const float* d, const float* cdf, int n, const float* u_final,
                                                   int n_final, float* out) {
    for (int j = threadIdx.x; j < n_final; j += 64) out[j] = invert_cdf_at(d, cdf, n, u_final[j]);
}

// Round 0: bound check of the initial 512 uniform samples at the net's beta (volsdf.py:159-177).
// Converged rays get their 64 fine samples now; the others are queued with beta+ = beta+_0.
__global__ void __launch_bounds__(64)
k_first_check(SamplerParams P, const float* __restrict__ dA, const float* __restrict__ sA,
              const float* __restrict__ u_final, float beta_plus0_denom, const float* __restrict__ far,
              float far_s, float* __restrict__ d_fine, float* __restrict__ beta_plus, float* __restrict__ beta_map,
              float* __restrict__ iter_usage, int* __restrict__ act_out, int* __restrict__ act_count) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ray = blockIdx.x;
    float* d = sm; float* s = sm + P.n; float* cdf = sm + 2 * P.n;
    load_row(d, dA + (size_t)ray * P.cap, P.n);
    load_row(s, sA + (size_t)ray * P.cap, P.n);
    __syncthreads();
    const float mx = error_bound_scan(d, s, P.n, P.alpha_net, P.beta_net, nullptr, false);
    const float fr = far ? far[ray] : far_s;
    if (in_guard_band(P, mx)) { escalate(P, ray); return; }
    if (!(mx > P.eps)) {
        opacity_cdf(d, s, P.n, P.alpha_net, P.beta_net, cdf);
        __syncthreads();
        emit_final_samples(d, cdf, P.n, u_final + (size_t)ray * P.u_final_stride, P.n_final, d_fine + (size_t)ray * P.n_final);
        if (threadIdx.x == 0) { iter_usage[ray] = 0.f; beta_map[ray] = P.beta_net; }
    } else if (threadIdx.x == 0) {
        beta_plus[ray] = sqrtf((fr * fr) / beta_plus0_denom);           // volsdf.py:149
        act_out[atomicAdd(act_count, 1)] = ray;
    }
}

// Up-sample an active ray: 512 new depths by inverting the CDF of the current error bound
// (sample_pdf(d, bounds, N_up + 2, det=True)[1:-1], volsdf.py:196), sorted.
__global__ void __launch_bounds__(64)
k_upsample(SamplerParams P, const float* __restrict__ dA, const float* __restrict__ sA,
           const int* __restrict__ act, const float* __restrict__ beta_plus, const float* __restrict__ u_up,
           int clamp_bounds, float* __restrict__ d_new) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int slot = blockIdx.x, ray = act[slot];
    const int n = P.n;
    float* d = sm; float* s = sm + n; float* cdf = sm + 2 * n; float* w = sm + 3 * n; float* out = sm + 4 * n;  // out: n_up
    load_row(d, dA + (size_t)ray * P.cap, n);
    load_row(s, sA + (size_t)ray * P.cap, n);
    __syncthreads();
    const float bp = beta_plus[ray];
    error_bound_scan(d, s, n, 1.f / bp, bp, w, clamp_bounds != 0);
    __syncthreads();
    // pdf = (w + 1e-5) / sum; cdf = [0, cumsum(pdf)]   (rend_util.py:260-265)
    // (round 6 b, measured and not kept: the bounds, the pdf and the running cdf of a lane's segment in registers instead of three passes over the LDS
    // row w - 264 VGPRs at 16 intervals per lane, spills at 24: one wave per SIMD where this kernel, which waits on dependent binary-search reads, has two)
    const int lane = threadIdx.x;
    const int nint = n - 1, seg = (nint + 63) >> 6, k0 = lane * seg, k1 = (k0 + seg < nint) ? k0 + seg : nint;
    float part = 0.f;
    for (int k = k0; k < k1; ++k) { const float wk = w[k] + 1e-5f; w[k] = wk; part += wk; }
    const float total = wave_sum(part);
    float ps = 0.f;
    for (int k = k0; k < k1; ++k) ps += w[k] / total;
    float run = wave_excl_sum(ps);
    if (lane == 0) cdf[0] = 0.f;
    for (int k = k0; k < k1; ++k) { run += w[k] / total; cdf[k + 1] = run; }
    __syncthreads();
    for (int j = lane; j < P.n_up; j += 64) out[j] = invert_cdf_at(d, cdf, n, u_up[j + 1]);
    bitonic_sort(out, P.n_up);
    for (int j = lane; j < P.n_up; j += 64) d_new[(size_t)slot * P.n_up + j] = out[j];
}

// What k_merge_check does once the merged rows are in LDS: the bound check at the net's beta, then sampling (converged) or the bisection for beta+.
// MAXSEG > 0: the lane's beta-independent interval constants are loaded once and every scan runs from registers (ray_common.h::SegConsts);
// MAXSEG = 0: every scan through error_bound_scan (any row length; NERFART_SCAN_GENERIC builds this form everywhere).
template <int MAXSEG>
__device__ __forceinline__ void merge_check_tail(const SamplerParams& P, int ray, const float* d, const float* s, float* cdf, int nm,
                                                 const float* __restrict__ u_final, float* __restrict__ d_fine, float* __restrict__ beta_plus,
                                                 float* __restrict__ beta_map, float* __restrict__ iter_usage, int* __restrict__ act_out,
                                                 int* __restrict__ act_count) {
    constexpr int NS = MAXSEG > 0 ? MAXSEG : 1;
    SegConsts<NS> C;
    if constexpr (MAXSEG > 0) load_seg_consts<NS>(d, s, nm, C);
    auto scan = [&](float alpha, float beta) -> float {
        if constexpr (MAXSEG > 0) return scan_consts<NS>(C, alpha, beta);
        else return error_bound_scan(d, s, nm, alpha, beta, nullptr, false);
    };
    const float mx = scan(P.alpha_net, P.beta_net);
    if (in_guard_band(P, mx)) { escalate(P, ray); return; }
    if (!(mx > P.eps)) {
        // cdf = the two depth rows of the merge: dead by now, exactly n + nu = nm floats
        opacity_cdf(d, s, nm, P.alpha_net, P.beta_net, cdf);
        __syncthreads();
        emit_final_samples(d, cdf, nm, u_final + (size_t)ray * P.u_final_stride, P.n_final, d_fine + (size_t)ray * P.n_final);
        if (threadIdx.x == 0) { iter_usage[ray] = (float)P.it; beta_map[ray] = P.beta_net; }
    } else {
        float hi = beta_plus[ray], lo = P.beta_net;
        for (int b = 0; b < P.max_bisect; ++b) {
            const float mid = 0.5f * (lo + hi);
            const float m = scan(1.f / mid, mid);
            if (m <= P.eps) hi = mid; else lo = mid;
        }
        if (threadIdx.x == 0) {
            beta_plus[ray] = hi;
            act_out[atomicAdd(act_count, 1)] = ray;
        }
    }
}

// Merge the 512 new (depth, sdf) pairs into the ray's sorted sample set, re-check the bound at the
// net's beta; converged -> 64 fine samples; else bisection for beta+ and re-queue (volsdf.py:211-285).
__global__ void __launch_bounds__(64)
k_merge_check(SamplerParams P, const float* __restrict__ dA, const float* __restrict__ sA,
              float* __restrict__ dB, float* __restrict__ sB, const int* __restrict__ act,
              const float* __restrict__ d_new, const float* __restrict__ s_new, const float* __restrict__ u_final,
              float* __restrict__ d_fine, float* __restrict__ beta_plus, float* __restrict__ beta_map,
              float* __restrict__ iter_usage, int* __restrict__ act_out, int* __restrict__ act_count) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int slot = blockIdx.x, ray = act[slot];
    const int n = P.n, nu = P.n_up, nm = n + nu;
    // LDS: the two depth rows (searched), then the merged rows; the sdf values are read once each, straight from global memory
    // (round 3: 3 (n + n_up) floats instead of 4 - one more ray per CU in every round; results unchanged)
    float* d_old = sm; float* dn = sm + n;
    float* d = dn + nu; float* s = d + nm;       // merged
    const float* s_old = sA + (size_t)ray * P.cap;
    const float* sn = s_new + (size_t)slot * nu;
    load_row(d_old, dA + (size_t)ray * P.cap, n);
    load_row(dn, d_new + (size_t)slot * nu, nu);
    __syncthreads();
    // stable merge by rank: old element i goes to i + #{new < d_old[i]}; new element k to k + #{old <= d_new[k]}
    for (int i = threadIdx.x; i < n; i += 64) {
        const int r = i + lower_bound(dn, nu, d_old[i]);
        d[r] = d_old[i]; s[r] = s_old[i];
    }
    for (int k = threadIdx.x; k < nu; k += 64) {
        const int r = k + upper_bound(d_old, n, dn[k]);
        d[r] = dn[k]; s[r] = sn[k];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nm; i += 64) {
        dB[(size_t)ray * P.cap + i] = d[i];
        sB[(size_t)ray * P.cap + i] = s[i];
    }
#ifndef NERFART_SCAN_GENERIC
    // rounds 1 and 2 (1,024 / 1,536 merged samples: every undecided ray of a frame / 78 % of them): the check and the 10 bisection scans from registers
    const int seg = (nm - 1 + 63) >> 6;
    if (seg <= 16) { merge_check_tail<16>(P, ray, d, s, d_old, nm, u_final, d_fine, beta_plus, beta_map, iter_usage, act_out, act_count); return; }
    if (seg <= 24) { merge_check_tail<24>(P, ray, d, s, d_old, nm, u_final, d_fine, beta_plus, beta_map, iter_usage, act_out, act_count); return; }
#endif
    merge_check_tail<0>(P, ray, d, s, d_old, nm, u_final, d_fine, beta_plus, beta_map, iter_usage, act_out, act_count);
}

// Rays still active after the last round: sample with the last beta+ (volsdf.py:294-300).
__global__ void __launch_bounds__(64)
k_finalize_unconverged(SamplerParams P, const float* __restrict__ dA, const float* __restrict__ sA,
                       const int* __restrict__ act, const float* __restrict__ u_final,
                       const float* __restrict__ beta_plus, float* __restrict__ d_fine,
                       float* __restrict__ beta_map, float* __restrict__ iter_usage) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ray = act[blockIdx.x];
    const int n = P.n;
    float* d = sm; float* s = sm + n; float* cdf = sm + 2 * n;
    load_row(d, dA + (size_t)ray * P.cap, n);
    load_row(s, sA + (size_t)ray * P.cap, n);
    __syncthreads();
    const float bp = beta_plus[ray];
    opacity_cdf(d, s, n, 1.f / bp, bp, cdf);
    __syncthreads();
    emit_final_samples(d, cdf, n, u_final + (size_t)ray * P.u_final_stride, P.n_final, d_fine + (size_t)ray * P.n_final);
    if (threadIdx.x == 0) { iter_usage[ray] = -1.f; beta_map[ray] = bp; }
}

// d_all = sort(cat(d_coarse, d_fine))  (volsdf.py:501-502); npad = next power of two >= na + nb
__global__ void __launch_bounds__(64)
k_sort_concat(const float* __restrict__ a, int na, int a_stride, const float* __restrict__ b, int nb, int b_stride,
              int npad, float* __restrict__ out, int out_stride) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ray = blockIdx.x;
    for (int i = threadIdx.x; i < npad; i += 64) {
        float v = INFINITY;
        if (i < na) v = a[(size_t)ray * a_stride + i];
        else if (i < na + nb) v = b[(size_t)ray * b_stride + (i - na)];
        sm[i] = v;
    }
    bitonic_sort(sm, npad);
    for (int i = threadIdx.x; i < na + nb; i += 64) out[(size_t)ray * out_stride + i] = sm[i];
}

// a16 compositing (volsdf.py:544-576): one wave per ray, P points -> P-1 intervals.
__global__ void __launch_bounds__(64)
k_composite_volsdf(int P, const float* __restrict__ d_all, const float* __restrict__ sdf,
                   const float* __restrict__ radiance, const float* __restrict__ nabla, float alpha, float beta,
                   int white_bkgd, float* __restrict__ rgb, float* __restrict__ depth, float* __restrict__ acc,
                   float* __restrict__ normals, float* __restrict__ sigma_out, float* __restrict__ p_out,
                   float* __restrict__ tau_out) {
    const int ray = blockIdx.x, lane = threadIdx.x;
    const int nint = P - 1;
    const int seg = (nint + 63) >> 6;
    const int k0 = lane * seg, k1 = (k0 + seg < nint) ? k0 + seg : nint;
    const float* dr = d_all + (size_t)ray * P;
    const float* sr = sdf + (size_t)ray * P;
    // pass 1: local product of p_i
    float lp = 1.f;
    for (int k = k0; k < k1; ++k) {
        const float sg = sdf_to_sigma(sr[k], alpha, beta);
        lp *= expf(-fmaxf(sg * (dr[k + 1] - dr[k]), 0.f));
    }
    float T = wave_excl_prod(lp);
    float r = 0.f, g = 0.f, b = 0.f, a = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
    for (int k = k0; k < k1; ++k) {
        const float sg = sdf_to_sigma(sr[k], alpha, beta);
        const float p = expf(-fmaxf(sg * (dr[k + 1] - dr[k]), 0.f));
        const float tau = (1.f - p + 1e-10f) * T;
        const size_t q = (size_t)ray * P + k;
        r += tau * radiance[3 * q]; g += tau * radiance[3 * q + 1]; b += tau * radiance[3 * q + 2];
        a += tau;
        if (normals) {
            const float vx = nabla[3 * q], vy = nabla[3 * q + 1], vz = nabla[3 * q + 2];
            const float nr = fmaxf(sqrtf(vx * vx + vy * vy + vz * vz), 1e-12f);     // F.normalize eps
            nx += vx / nr * tau; ny += vy / nr * tau; nz += vz / nr * tau;
        }
        if (sigma_out) sigma_out[q] = sg;
        if (p_out) p_out[(size_t)ray * nint + k] = p;
        if (tau_out) tau_out[(size_t)ray * nint + k] = tau;
        T *= p;
    }
    if (sigma_out && lane == 0) sigma_out[(size_t)ray * P + P - 1] = sdf_to_sigma(sr[P - 1], alpha, beta);
    r = wave_sum(r); g = wave_sum(g); b = wave_sum(b); a = wave_sum(a);
    // depth = sum tau / (sum tau + 1e-10) * d   (second pass needs the total)
    T = wave_excl_prod(lp);
    float dp = 0.f;
    const float inv = a + 1e-10f;
    for (int k = k0; k < k1; ++k) {
        const float sg = sdf_to_sigma(sr[k], alpha, beta);
        const float p = expf(-fmaxf(sg * (dr[k + 1] - dr[k]), 0.f));
        const float tau = (1.f - p + 1e-10f) * T;
        dp += tau / inv * dr[k];
        T *= p;
    }
    dp = wave_sum(dp);
    if (normals) { nx = wave_sum(nx); ny = wave_sum(ny); nz = wave_sum(nz); }
    if (lane == 0) {
        if (white_bkgd) { r += 1.f - a; g += 1.f - a; b += 1.f - a; }
        rgb[3 * (size_t)ray] = r; rgb[3 * (size_t)ray + 1] = g; rgb[3 * (size_t)ray + 2] = b;
        depth[ray] = dp; acc[ray] = a;
        if (normals) { normals[3 * (size_t)ray] = nx; normals[3 * (size_t)ray + 1] = ny; normals[3 * (size_t)ray + 2] = nz; }
    }
}

// ---- escalation of guard-band / never-converged rays (fine_sample_run): compact the listed rays, run Algorithm 1 on them again, scatter --------
__global__ void k_gather_rays(const int* __restrict__ list, int n, const float* __restrict__ rays_o, const float* __restrict__ rays_dn,
                              const float* __restrict__ near, const float* __restrict__ far, const float* __restrict__ u_final, int n_final,
                              float* __restrict__ c_o, float* __restrict__ c_dn, float* __restrict__ c_near, float* __restrict__ c_far,
                              float* __restrict__ c_u) {
    const int slot = blockIdx.x, ray = list[slot], t = threadIdx.x;
    if (slot >= n) return;
    if (t < 3) { c_o[3 * slot + t] = rays_o[3 * (size_t)ray + t]; c_dn[3 * slot + t] = rays_dn[3 * (size_t)ray + t]; }
    if (t == 0 && near) c_near[slot] = near[ray];
    if (t == 0 && far) c_far[slot] = far[ray];
    if (u_final) for (int j = t; j < n_final; j += 64) c_u[(size_t)slot * n_final + j] = u_final[(size_t)ray * n_final + j];
}

__global__ void k_scatter_samples(const int* __restrict__ list, int n, int n_final, const float* __restrict__ c_d_fine,
                                  const float* __restrict__ c_beta, const float* __restrict__ c_iter, float* __restrict__ d_fine,
                                  float* __restrict__ beta_map, float* __restrict__ iter_usage) {
    const int slot = blockIdx.x, ray = list[slot], t = threadIdx.x;
    if (slot >= n) return;
    for (int j = t; j < n_final; j += 64) d_fine[(size_t)ray * n_final + j] = c_d_fine[(size_t)slot * n_final + j];
    if (t == 0) { beta_map[ray] = c_beta[slot]; iter_usage[ray] = c_iter[slot]; }
}

static inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }
static inline int next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }

}  // namespace nerfart

using namespace nerfart;

extern "C" {

// ---- forward declarations of the MLP entry points (mlp_chain.hip) ---------------------
int nerfart_sdf_fwd_rays(const float*, int, const float*, const float*, const int*, const float*, int, int, int, float, float*, int, void*);
int nerfart_sdf_nabla_fwd_rays(const float*, int, const float*, const float*, const int*, const float*, int, int, int, float, float*, float*, float*, void*,
                               long long, void*);
long long nerfart_sdf_nabla_workspace_bytes(int precision);
int nerfart_radiance_fwd_rays(const float*, int, int, const float*, const float*, const int*, const float*, int, int, int, const float*, const float*, float*, void*);

// torch.linspace(start, end, n) in fp32: step = (end-start)/(n-1); the first half counts up from
// start, the second half down from end (ATen RangeFactories linspace kernel).  Host helper.
void nerfart_linspace(float start, float end, int n, float* out) {
    if (n == 1) { out[0] = start; return; }
    const float step = (end - start) / (float)(n - 1);
    const int half = n / 2;
    for (int i = 0; i < n; ++i) out[i] = (i < half) ? start + step * (float)i : end - step * (float)(n - i - 1);
}

// ---- stage entry points (also used one by one by the parity tests) --------------------
int nerfart_normalize_dirs(const float* in, float* out, int n, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_normalize_dirs, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, in, out, n);
    NERFART_HIP(hipGetLastError());
    return 0;
}

int nerfart_linspace_depths(const float* t_dev, int n, const float* near, const float* far, float near_s, float far_s,
                            int n_rays, float* out, int stride, void* stream) {
    const long long tot = (long long)n_rays * n;
    if (tot <= 0) return 0;
    hipLaunchKernelGGL(k_linspace_depths, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       t_dev, n, near, far, near_s, far_s, n_rays, out, stride);
    NERFART_HIP(hipGetLastError());
    return 0;
}

static int set_lds(const void* k, size_t bytes) {
    if (bytes > 160 * 1024) { set_last_error("per-ray kernel needs more than 160 KiB of LDS (too many samples per ray)"); return 2; }
    NERFART_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

// guard / esc_list / esc_count: see SamplerParams (the exported stage entry point runs with the guard off)
static int first_check_launch(int n_rays, int n, int cap, int n_final, float eps, float alpha_net, float beta_net,
                              const float* dA, const float* sA, const float* u_final, int u_final_stride,
                              float beta_plus0_denom, const float* far, float far_s, float* d_fine, float* beta_plus,
                              float* beta_map, float* iter_usage, int* act_out, int* act_count, float guard, int* esc_list, int* esc_count,
                              void* stream) {
    if (n_rays <= 0) return 0;
    SamplerParams P{n, cap, 0, n_final, 0, 0, eps, alpha_net, beta_net, u_final_stride, guard, esc_list, esc_count};
    const size_t lds = (size_t)3 * n * sizeof(float);
    if (int rc = set_lds((const void*)k_first_check, lds)) return rc;
    hipLaunchKernelGGL(k_first_check, dim3(n_rays), dim3(64), lds, (hipStream_t)stream, P, dA, sA, u_final,
                       beta_plus0_denom, far, far_s, d_fine, beta_plus, beta_map, iter_usage, act_out, act_count);
    NERFART_HIP(hipGetLastError());
    return 0;
}

int nerfart_volsdf_first_check(int n_rays, int n, int cap, int n_final, float eps, float alpha_net, float beta_net,
                               const float* dA, const float* sA, const float* u_final, int u_final_stride,
                               float beta_plus0_denom, const float* far, float far_s, float* d_fine, float* beta_plus,
                               float* beta_map, float* iter_usage, int* act_out, int* act_count, void* stream) {
    return first_check_launch(n_rays, n, cap, n_final, eps, alpha_net, beta_net, dA, sA, u_final, u_final_stride, beta_plus0_denom, far, far_s, d_fine,
                              beta_plus, beta_map, iter_usage, act_out, act_count, 0.f, nullptr, nullptr, stream);
}

int nerfart_volsdf_upsample(int n_active, int n, int cap, int n_up, const float* dA, const float* sA, const int* act,
                            const float* beta_plus, const float* u_up, int clamp_bounds, float* d_new, void* stream) {
    if (n_active <= 0) return 0;
    if (n_up & (n_up - 1)) { set_last_error("n_up must be a power of two"); return 2; }
    SamplerParams P{n, cap, n_up, 0, 0, 0, 0.f, 0.f, 0.f};
    const size_t lds = ((size_t)4 * n + n_up) * sizeof(float);
    if (int rc = set_lds((const void*)k_upsample, lds)) return rc;
    hipLaunchKernelGGL(k_upsample, dim3(n_active), dim3(64), lds, (hipStream_t)stream, P, dA, sA, act, beta_plus, u_up,
                       clamp_bounds, d_new);
    NERFART_HIP(hipGetLastError());
    return 0;
}

static int merge_check_launch(int n_active, int n, int cap, int n_up, int n_final, int max_bisect, int it, float eps,
                              float alpha_net, float beta_net, const float* dA, const float* sA, float* dB, float* sB,
                              const int* act, const float* d_new, const float* s_new, const float* u_final,
                              int u_final_stride, float* d_fine, float* beta_plus, float* beta_map, float* iter_usage,
                              int* act_out, int* act_count, float guard, int* esc_list, int* esc_count, void* stream) {
    if (n_active <= 0) return 0;
    SamplerParams P{n, cap, n_up, n_final, max_bisect, it, eps, alpha_net, beta_net, u_final_stride, guard, esc_list, esc_count};
    const size_t lds = ((size_t)3 * n + 3 * n_up) * sizeof(float);
    if (int rc = set_lds((const void*)k_merge_check, lds)) return rc;
    hipLaunchKernelGGL(k_merge_check, dim3(n_active), dim3(64), lds, (hipStream_t)stream, P, dA, sA, dB, sB, act, d_new,
                       s_new, u_final, d_fine, beta_plus, beta_map, iter_usage, act_out, act_count);
    NERFART_HIP(hipGetLastError());
    return 0;
}

int nerfart_volsdf_merge_check(int n_active, int n, int cap, int n_up, int n_final, int max_bisect, int it, float eps,
                               float alpha_net, float beta_net, const float* dA, const float* sA, float* dB, float* sB,
                               const int* act, const float* d_new, const float* s_new, const float* u_final,
                               int u_final_stride, float* d_fine, float* beta_plus, float* beta_map, float* iter_usage,
                               int* act_out, int* act_count, void* stream) {
    return merge_check_launch(n_active, n, cap, n_up, n_final, max_bisect, it, eps, alpha_net, beta_net, dA, sA, dB, sB, act, d_new, s_new, u_final,
                              u_final_stride, d_fine, beta_plus, beta_map, iter_usage, act_out, act_count, 0.f, nullptr, nullptr, stream);
}

int nerfart_volsdf_finalize(int n_active, int n, int cap, int n_final, const float* dA, const float* sA, const int* act,
                            const float* u_final, int u_final_stride, const float* beta_plus, float* d_fine,
                            float* beta_map, float* iter_usage, void* stream) {
    if (n_active <= 0) return 0;
    SamplerParams P{n, cap, 0, n_final, 0, 0, 0.f, 0.f, 0.f, u_final_stride};
    const size_t lds = (size_t)3 * n * sizeof(float);
    if (int rc = set_lds((const void*)k_finalize_unconverged, lds)) return rc;
    hipLaunchKernelGGL(k_finalize_unconverged, dim3(n_active), dim3(64), lds, (hipStream_t)stream, P, dA, sA, act,
                       u_final, beta_plus, d_fine, beta_map, iter_usage);
    NERFART_HIP(hipGetLastError());
    return 0;
}

int nerfart_sort_concat(int n_rays, const float* a, int na, int a_stride, const float* b, int nb, int b_stride,
                        float* out, int out_stride, void* stream) {
    if (n_rays <= 0) return 0;
    const int npad = next_pow2(na + nb);
    const size_t lds = (size_t)npad * sizeof(float);
    if (int rc = set_lds((const void*)k_sort_concat, lds)) return rc;
    hipLaunchKernelGGL(k_sort_concat, dim3(n_rays), dim3(64), lds, (hipStream_t)stream, a, na, a_stride, b, nb, b_stride,
                       npad, out, out_stride);
    NERFART_HIP(hipGetLastError());
    return 0;
}

int nerfart_volsdf_composite(int n_rays, int P, const float* d_all, const float* sdf, const float* radiance,
                             const float* nabla, float alpha, float beta, int white_bkgd, float* rgb, float* depth,
                             float* acc, float* normals, float* sigma_out, float* p_out, float* tau_out, void* stream) {
    if (n_rays <= 0) return 0;
    if (normals && !nabla) { set_last_error("composite: normals requested without nablas"); return 2; }
    hipLaunchKernelGGL(k_composite_volsdf, dim3(n_rays), dim3(64), 0, (hipStream_t)stream, P, d_all, sdf, radiance, nabla,
                       alpha, beta, white_bkgd, rgb, depth, acc, normals, sigma_out, p_out, tau_out);
    NERFART_HIP(hipGetLastError());
    return 0;
}

// ---- fine_sample (Algorithm 1) on the device --------------------------------------------
// Workspace (caller-allocated, device): see nerfart_volsdf_sampler_workspace_bytes.
typedef struct {
    float *dA, *sA, *dB, *sB, *d_new, *s_new, *beta_plus;
    int *act0, *act1, *count;
    float *t_init, *u_up, *u_final;
    // escalation (behind everything a run on <= n_rays rays carves, so the second run may reuse the front of the same workspace)
    int* esc_list;
    float *c_o, *c_dn, *c_near, *c_far, *c_u, *c_d_fine, *c_beta, *c_iter;
} sampler_ws_t;

enum { COUNT_SLOTS = 64, ESC_SLOT = COUNT_SLOTS - 1 };     // w.count: one active-ray counter per round + the escalation list's length

static size_t carve_sampler(char* base, int R, int cap, int n_up, int n0, int n_final, sampler_ws_t* w) {
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += align_up(bytes); return base ? base + r : (char*)nullptr; };
    const size_t row = (size_t)R * cap * sizeof(float);
    float* p;
    p = (float*)take(row); if (w) w->dA = p;
    p = (float*)take(row); if (w) w->sA = p;
    p = (float*)take(row); if (w) w->dB = p;
    p = (float*)take(row); if (w) w->sB = p;
    p = (float*)take((size_t)R * n_up * sizeof(float)); if (w) w->d_new = p;
    p = (float*)take((size_t)R * n_up * sizeof(float)); if (w) w->s_new = p;
    p = (float*)take((size_t)R * sizeof(float)); if (w) w->beta_plus = p;
    int* q;
    q = (int*)take((size_t)R * sizeof(int)); if (w) w->act0 = q;
    q = (int*)take((size_t)R * sizeof(int)); if (w) w->act1 = q;
    q = (int*)take(COUNT_SLOTS * sizeof(int)); if (w) w->count = q;
    p = (float*)take((size_t)n0 * sizeof(float)); if (w) w->t_init = p;
    p = (float*)take((size_t)(n_up + 2) * sizeof(float)); if (w) w->u_up = p;
    p = (float*)take((size_t)n_final * sizeof(float)); if (w) w->u_final = p;
    q = (int*)take((size_t)R * sizeof(int)); if (w) w->esc_list = q;
    p = (float*)take((size_t)R * 3 * sizeof(float)); if (w) w->c_o = p;
    p = (float*)take((size_t)R * 3 * sizeof(float)); if (w) w->c_dn = p;
    p = (float*)take((size_t)R * sizeof(float)); if (w) w->c_near = p;
    p = (float*)take((size_t)R * sizeof(float)); if (w) w->c_far = p;
    p = (float*)take((size_t)R * n_final * sizeof(float)); if (w) w->c_u = p;
    p = (float*)take((size_t)R * n_final * sizeof(float)); if (w) w->c_d_fine = p;
    p = (float*)take((size_t)R * sizeof(float)); if (w) w->c_beta = p;
    p = (float*)take((size_t)R * sizeof(float)); if (w) w->c_iter = p;
    return o;
}

long long nerfart_volsdf_sampler_workspace_bytes(int n_rays, int n_init, int n_up, int n_final, int max_iter) {
    return (long long)carve_sampler(nullptr, n_rays, n_init + max_iter * n_up, n_up, n_init, n_final, nullptr);
}

// fine_sample (volsdf.py:97-302) for n_rays rays with already normalised directions.
//   near/far: per-ray device arrays or nullptr + scalars.  Outputs: d_fine [R, n_final],
//   beta_map [R], iter_usage [R] (float: 0..max_iter, -1 = never converged).
// GUARDED form (esc_blob != nullptr && guard > 0; nerfart_volsdf_fine_sample_guarded): the SDF queries run on (surf_blob, precision) - the cheap
// arithmetic - but every ray whose outcome hangs on a marginal threshold decision is sampled AGAIN, from its first query on, on (esc_blob,
// esc_precision):  (i) a ray whose max B lies within guard * eps of eps at a convergence check (volsdf.py:162-163, :240-242) stops there;
// (ii) a ray still active after the last round (volsdf.py:294-300: sampled with its last bisected beta+, the rays any change of rounding moves).
// (iii) late_round > 0 (ABI 5): a ray still active after round `late_round` - from there on every further round starts from a 10-step bisection for
// beta+ whose threshold decisions (volsdf.py:266-275) feed the next round's sampling density: the branch-sensitive rays of Algorithm 1, 3.6 % of a
// frame at late_round = 3 - is escalated there instead of being carried through the remaining rounds on the cheap arithmetic.
// Rays are independent, so the escalated rays' samples are bit-identical to a run of the whole batch on esc_blob.
static int fine_sample_run(const float* surf_blob, int precision, const float* esc_blob, int esc_precision, float guard, int late_round,
                           const float* rays_o, const float* rays_dn, int n_rays,
                           const float* near, const float* far, float near_s, float far_s, float R_bg,
                           float alpha_net, float beta_net, float eps, int n_init, int n_up, int n_final,
                           int max_iter, int max_bisect, const float* t_init_dev, const float* u_up_dev,
                           const float* u_final_dev, int u_final_per_ray, float* d_fine, float* beta_map,
                           float* iter_usage, void* workspace, long long workspace_bytes, int* n_escalated, hipStream_t stream) {
    if (n_escalated) *n_escalated = 0;
    if (n_rays <= 0) return 0;
    if (u_final_per_ray && !u_final_dev) { set_last_error("fine_sample: u_final_per_ray needs u_final_dev [n_rays, n_final]"); return 2; }
    if (max_iter < 0 || max_iter >= ESC_SLOT) { set_last_error("fine_sample: max_iter must be in [0, 62]"); return 2; }
    const bool guarded = esc_blob != nullptr && guard > 0.f;
    if (!guarded) guard = 0.f;
    const int u_stride = u_final_per_ray ? n_final : 0;
    const int cap = n_init + max_iter * n_up;
    sampler_ws_t w;
    const size_t need = carve_sampler((char*)workspace, n_rays, cap, n_up, n_init, n_final, &w);
    if (!workspace || (size_t)workspace_bytes < need) { set_last_error("fine_sample: workspace too small"); return 2; }
    // linspace tables: torch.linspace(0, 1, n) for n = n_init, n_up + 2, n_final.  Callers that hold the
    // host framework's own tables pass them (torch's CPU kernel is vectorised and differs from the scalar
    // formula by an ulp on some entries); otherwise the library's nerfart_linspace is used.
    const bool own_tables = !(t_init_dev && u_up_dev && u_final_dev);
    if (!own_tables) {
        w.t_init = const_cast<float*>(t_init_dev); w.u_up = const_cast<float*>(u_up_dev); w.u_final = const_cast<float*>(u_final_dev);
    } else {
        float* h = (float*)malloc(sizeof(float) * (size_t)(n_init + n_up + 2 + n_final));
        if (!h) { set_last_error("out of host memory"); return 3; }
        nerfart_linspace(0.f, 1.f, n_init, h);
        nerfart_linspace(0.f, 1.f, n_up + 2, h + n_init);
        nerfart_linspace(0.f, 1.f, n_final, h + n_init + n_up + 2);
        hipError_t e1 = hipMemcpyAsync(w.t_init, h, sizeof(float) * n_init, hipMemcpyHostToDevice, stream);
        hipError_t e2 = hipMemcpyAsync(w.u_up, h + n_init, sizeof(float) * (n_up + 2), hipMemcpyHostToDevice, stream);
        hipError_t e3 = hipMemcpyAsync(w.u_final, h + n_init + n_up + 2, sizeof(float) * n_final, hipMemcpyHostToDevice, stream);
        hipError_t e4 = hipStreamSynchronize(stream);
        free(h);
        NERFART_HIP(e1); NERFART_HIP(e2); NERFART_HIP(e3); NERFART_HIP(e4);
        if (u_final_per_ray) w.u_final = const_cast<float*>(u_final_dev);
    }
    if (int rc = nerfart_linspace_depths(w.t_init, n_init, near, far, near_s, far_s, n_rays, w.dA, cap, stream)) return rc;
    if (int rc = nerfart_sdf_fwd_rays(surf_blob, precision, rays_o, rays_dn, nullptr, w.dA, n_rays, n_init, cap, R_bg, w.sA, cap, stream)) return rc;
    NERFART_HIP(hipMemsetAsync(w.count, 0, COUNT_SLOTS * sizeof(int), stream));
    const float denom = (float)(4.0 * (double)(n_init - 1) * log(1.0 + (double)eps));     // volsdf.py:149
    if (int rc = first_check_launch(n_rays, n_init, cap, n_final, eps, alpha_net, beta_net, w.dA, w.sA, w.u_final,
                                    u_stride, denom, far, far_s, d_fine, w.beta_plus, beta_map, iter_usage, w.act0,
                                    w.count, guard, w.esc_list, w.count + ESC_SLOT, stream)) return rc;
    // one read of the counter block per round (the active count of the round + the escalation list's length so far): the only host synchronisation
    int h_count[COUNT_SLOTS];
    NERFART_HIP(hipMemcpyAsync(h_count, w.count, sizeof(h_count), hipMemcpyDeviceToHost, stream));
    NERFART_HIP(hipStreamSynchronize(stream));
    int n_act = h_count[0];
    float *dA = w.dA, *sA = w.sA, *dB = w.dB, *sB = w.sB;
    int *act = w.act0, *act_next = w.act1;
    int n = n_init;
    for (int it = 1; it <= max_iter && n_act > 0; ++it) {
        if (int rc = nerfart_volsdf_upsample(n_act, n, cap, n_up, dA, sA, act, w.beta_plus, w.u_up, it > 1, w.d_new, stream)) return rc;
        if (int rc = nerfart_sdf_fwd_rays(surf_blob, precision, rays_o, rays_dn, act, w.d_new, n_act, n_up, n_up, R_bg, w.s_new, n_up, stream)) return rc;
        if (int rc = merge_check_launch(n_act, n, cap, n_up, n_final, max_bisect, it, eps, alpha_net, beta_net, dA, sA,
                                        dB, sB, act, w.d_new, w.s_new, w.u_final, u_stride, d_fine, w.beta_plus,
                                        beta_map, iter_usage, act_next, w.count + it, guard, w.esc_list, w.count + ESC_SLOT, stream)) return rc;
        NERFART_HIP(hipMemcpyAsync(h_count, w.count, sizeof(h_count), hipMemcpyDeviceToHost, stream));
        NERFART_HIP(hipStreamSynchronize(stream));
        n_act = h_count[it];
        float* t;
        t = dA; dA = dB; dB = t;
        t = sA; sA = sB; sB = t;
        int* ti = act; act = act_next; act_next = ti;
        n += n_up;
        if (guarded && late_round > 0 && it >= late_round) break;       // the rays still active go to the escalation list below
    }
    if (!guarded) {
        if (n_act > 0)
            if (int rc = nerfart_volsdf_finalize(n_act, n, cap, n_final, dA, sA, act, w.u_final, u_stride, w.beta_plus, d_fine,
                                                 beta_map, iter_usage, stream)) return rc;
        return 0;
    }
    // ---- escalation: the guard-band rays listed so far + the rays that never converged -> Algorithm 1 again, on the escalation blob ----
    int n_esc = h_count[ESC_SLOT];
    if (n_act > 0) {
        NERFART_HIP(hipMemcpyAsync(w.esc_list + n_esc, act, sizeof(int) * (size_t)n_act, hipMemcpyDeviceToDevice, stream));
        n_esc += n_act;
    }
    if (n_escalated) *n_escalated = n_esc;
    if (n_esc == 0) return 0;
    hipLaunchKernelGGL(k_gather_rays, dim3(n_esc), dim3(64), 0, stream, w.esc_list, n_esc, rays_o, rays_dn, near, far,
                       u_final_per_ray ? u_final_dev : (const float*)nullptr, n_final, w.c_o, w.c_dn, w.c_near, w.c_far, w.c_u);
    NERFART_HIP(hipGetLastError());
    // the second run carves the FRONT of this workspace (its n_esc <= n_rays rays need no more than this run's, whose contents are dead); the
    // compacted rays, their outputs and the list sit behind it.  Tables the library built itself sit in the front too: the second run rebuilds them.
    const float* u2 = u_final_per_ray ? w.c_u : (own_tables ? nullptr : u_final_dev);
    const int* esc_list = w.esc_list;
    float *c_d_fine = w.c_d_fine, *c_beta = w.c_beta, *c_iter = w.c_iter;
    profile_class0_as(4);                     // launch profiling: the escalation run's SDF queries are their own class (nerfart_profile_end5)
    const int rc_esc = fine_sample_run(esc_blob, esc_precision, nullptr, 0, 0.f, 0, w.c_o, w.c_dn, n_esc, near ? w.c_near : nullptr, far ? w.c_far : nullptr,
                                       near_s, far_s, R_bg, alpha_net, beta_net, eps, n_init, n_up, n_final, max_iter, max_bisect,
                                       own_tables ? nullptr : t_init_dev, own_tables ? nullptr : u_up_dev, u2, u_final_per_ray, c_d_fine, c_beta, c_iter,
                                       workspace, workspace_bytes, nullptr, stream);
    profile_class0_as(0);
    if (rc_esc) return rc_esc;
    hipLaunchKernelGGL(k_scatter_samples, dim3(n_esc), dim3(64), 0, stream, esc_list, n_esc, n_final, c_d_fine, c_beta, c_iter, d_fine, beta_map,
                       iter_usage);
    NERFART_HIP(hipGetLastError());
    return 0;
}

int nerfart_volsdf_fine_sample(const float* surf_blob, int precision, const float* rays_o, const float* rays_dn, int n_rays,
                               const float* near, const float* far, float near_s, float far_s, float R_bg,
                               float alpha_net, float beta_net, float eps, int n_init, int n_up, int n_final,
                               int max_iter, int max_bisect, const float* t_init_dev, const float* u_up_dev,
                               const float* u_final_dev, int u_final_per_ray, float* d_fine, float* beta_map,
                               float* iter_usage, void* workspace, long long workspace_bytes, void* stream) {
    return fine_sample_run(surf_blob, precision, nullptr, 0, 0.f, 0, rays_o, rays_dn, n_rays, near, far, near_s, far_s, R_bg, alpha_net, beta_net, eps, n_init,
                           n_up, n_final, max_iter, max_bisect, t_init_dev, u_up_dev, u_final_dev, u_final_per_ray, d_fine, beta_map, iter_usage, workspace,
                           workspace_bytes, nullptr, (hipStream_t)stream);
}

int nerfart_volsdf_fine_sample_guarded2(const float* surf_blob, int precision, const float* esc_blob, int esc_precision, float guard, int late_round,
                                       const float* rays_o, const float* rays_dn, int n_rays,
                                       const float* near, const float* far, float near_s, float far_s, float R_bg,
                                       float alpha_net, float beta_net, float eps, int n_init, int n_up, int n_final,
                                       int max_iter, int max_bisect, const float* t_init_dev, const float* u_up_dev,
                                       const float* u_final_dev, int u_final_per_ray, float* d_fine, float* beta_map,
                                       float* iter_usage, int* n_escalated, void* workspace, long long workspace_bytes, void* stream) {
    return fine_sample_run(surf_blob, precision, esc_blob, esc_precision, guard, late_round, rays_o, rays_dn, n_rays, near, far, near_s, far_s, R_bg, alpha_net, beta_net, eps,
                           n_init, n_up, n_final, max_iter, max_bisect, t_init_dev, u_up_dev, u_final_dev, u_final_per_ray, d_fine, beta_map, iter_usage,
                           workspace, workspace_bytes, n_escalated, (hipStream_t)stream);
}

int nerfart_volsdf_fine_sample_guarded(const float* surf_blob, int precision, const float* esc_blob, int esc_precision, float guard,
                                       const float* rays_o, const float* rays_dn, int n_rays,
                                       const float* near, const float* far, float near_s, float far_s, float R_bg,
                                       float alpha_net, float beta_net, float eps, int n_init, int n_up, int n_final,
                                       int max_iter, int max_bisect, const float* t_init_dev, const float* u_up_dev,
                                       const float* u_final_dev, int u_final_per_ray, float* d_fine, float* beta_map,
                                       float* iter_usage, int* n_escalated, void* workspace, long long workspace_bytes, void* stream) {
    return nerfart_volsdf_fine_sample_guarded2(surf_blob, precision, esc_blob, esc_precision, guard, 0, rays_o, rays_dn, n_rays, near, far, near_s, far_s, R_bg,
                                               alpha_net, beta_net, eps, n_init, n_up, n_final, max_iter, max_bisect, t_init_dev, u_up_dev, u_final_dev,
                                               u_final_per_ray, d_fine, beta_map, iter_usage, n_escalated, workspace, workspace_bytes, stream);
}

// ---- whole-chunk VolSDF render (boundary B1: render_fn / volume_render, volsdf.py:389-615) ----
typedef struct {
    float *rays_dn, *d_fine, *d_coarse, *t_coarse, *d_all, *sdf, *nabla, *rad, *beta_map, *iter_usage, *h7;
    char* sampler;
    size_t sampler_bytes;
    char* nabla_ws;           // softplus' scratch of the reverse-mode grad(SDF) kernel (nerfart_sdf_nabla_workspace_bytes)
    size_t nabla_ws_bytes;
} render_ws_t;

static size_t carve_render(char* base, int R, int n_samples, int n_imp, int max_iter, int k3_rays, render_ws_t* w) {
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += align_up(bytes); return base ? base + r : (char*)nullptr; };
    const int P = n_samples + n_imp;
    const int n_init = 4 * n_samples, n_up = 4 * n_samples;
    float* p;
    p = (float*)take((size_t)R * 3 * 4); if (w) w->rays_dn = p;
    p = (float*)take((size_t)R * n_imp * 4); if (w) w->d_fine = p;
    p = (float*)take((size_t)R * n_samples * 4); if (w) w->d_coarse = p;
    p = (float*)take((size_t)n_samples * 4); if (w) w->t_coarse = p;
    p = (float*)take((size_t)R * P * 4); if (w) w->d_all = p;
    p = (float*)take((size_t)R * P * 4); if (w) w->sdf = p;
    p = (float*)take((size_t)R * P * 12); if (w) w->nabla = p;
    p = (float*)take((size_t)R * P * 12); if (w) w->rad = p;
    p = (float*)take((size_t)R * 4); if (w) w->beta_map = p;
    p = (float*)take((size_t)R * 4); if (w) w->iter_usage = p;
    const int rk = k3_rays < R ? k3_rays : R;
    p = (float*)take((size_t)rk * P * 256 * 4); if (w) w->h7 = p;
    const size_t sb = carve_sampler(nullptr, R, n_init + max_iter * n_up, n_up, n_init, n_imp, nullptr);
    char* sp = take(sb);
    if (w) { w->sampler = sp; w->sampler_bytes = sb; }
    const long long nb0 = nerfart_sdf_nabla_workspace_bytes(0), nb1 = nerfart_sdf_nabla_workspace_bytes(1);
    const size_t nb = (size_t)(nb0 > nb1 ? nb0 : nb1);          // sized for either precision
    char* np = take(nb);
    if (w) { w->nabla_ws = np; w->nabla_ws_bytes = nb; }
    return o;
}

long long nerfart_volsdf_render_workspace_bytes(int n_rays, int n_samples, int n_importance, int max_upsample_steps,
                                                int k3_rays_chunk) {
    return (long long)carve_render(nullptr, n_rays, n_samples, n_importance, max_upsample_steps, k3_rays_chunk, nullptr);
}

// Renders n_rays rays (rays_d un-normalised, as get_rays returns them).  Outputs rgb [R,3], depth [R],
// acc [R] always; every other output pointer may be null:  normals [R,3]; detailed per-sample
// arrays d_all/sdf/sigma [R,P], nabla/radiance [R,P,3], p_i/tau [R,P-1]; beta_map/iter_usage [R].
// Every stage on its own blob / precision (nerfart_volsdf_render_staged_fwd):
//   sampler_blob / sampler_precision: Algorithm 1's SDF queries (the no-gradient sampling stage, volsdf.py:479), with sampler_guard > 0: guarded -
//     rays whose convergence decision is marginal or that never converge are sampled again on (surf_blob, precision) (fine_sample_run);
//   surf_blob / precision: sdf + nabla of the 192 final samples;   rad_blob / rad_precision: the radiance net there;   compositing: fp32.
// nerfart_volsdf_render_mixed_fwd = this with rad_precision = precision and the guard off; nerfart_volsdf_render_fwd passes the same blob and
// precision everywhere.
int nerfart_volsdf_render_staged2_fwd(const float* surf_blob, int precision, const float* rad_blob, int rad_precision, const float* sampler_blob,
                                     int sampler_precision, float sampler_guard, int sampler_late_round, int view_tiles, const float* rays_o,
                              const float* rays_d, int n_rays, float near_s, float far_s, float R_bg, float alpha,
                              float beta, float eps, int n_samples, int n_importance, int max_upsample_steps,
                              int max_bisection_steps, int white_bkgd, int k3_rays_chunk, const float* t_coarse_dev,
                              const float* t_init_dev, const float* u_up_dev, const float* u_final_dev, int u_final_per_ray,
                              float* rgb, float* depth, float* acc, float* normals, float* d_all_out, float* sdf_out, float* nabla_out,
                              float* radiance_out, float* sigma_out, float* p_out, float* tau_out, float* beta_map_out,
                              float* iter_usage_out, int* n_escalated, void* workspace, long long workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rays <= 0) return 0;
    if (!sampler_blob) { set_last_error("render: sampler_blob is NULL"); return 2; }
    if (n_samples < 2 || n_importance < 1 || k3_rays_chunk < 1) { set_last_error("render: bad sample counts"); return 2; }
    const int P = n_samples + n_importance;
    render_ws_t w;
    const size_t need = carve_render((char*)workspace, n_rays, n_samples, n_importance, max_upsample_steps, k3_rays_chunk, &w);
    if (!workspace || (size_t)workspace_bytes < need) { set_last_error("render: workspace too small"); return 2; }
    float* d_all = d_all_out ? d_all_out : w.d_all;
    float* sdf = sdf_out ? sdf_out : w.sdf;
    float* nabla = nabla_out ? nabla_out : w.nabla;
    float* rad = radiance_out ? radiance_out : w.rad;
    float* beta_map = beta_map_out ? beta_map_out : w.beta_map;
    float* iter_usage = iter_usage_out ? iter_usage_out : w.iter_usage;

    if (int rc = nerfart_normalize_dirs(rays_d, w.rays_dn, n_rays, stream)) return rc;
    const bool guarded = sampler_guard > 0.f && (sampler_blob != surf_blob || sampler_precision != precision);
    if (int rc = fine_sample_run(sampler_blob, sampler_precision, guarded ? surf_blob : nullptr, precision, sampler_guard, sampler_late_round, rays_o, w.rays_dn, n_rays,
                                 nullptr, nullptr, near_s, far_s, R_bg, alpha, beta, eps, 4 * n_samples, 4 * n_samples, n_importance,
                                 max_upsample_steps, max_bisection_steps, t_init_dev, u_up_dev, u_final_dev,
                                 u_final_per_ray, w.d_fine, beta_map, iter_usage, w.sampler, (long long)w.sampler_bytes, n_escalated, stream)) return rc;
    if (t_coarse_dev) {
        w.t_coarse = const_cast<float*>(t_coarse_dev);
    } else {
        float* h = (float*)malloc(sizeof(float) * (size_t)n_samples);
        if (!h) { set_last_error("out of host memory"); return 3; }
        nerfart_linspace(0.f, 1.f, n_samples, h);
        hipError_t e1 = hipMemcpyAsync(w.t_coarse, h, sizeof(float) * n_samples, hipMemcpyHostToDevice, stream);
        hipError_t e2 = hipStreamSynchronize(stream);
        free(h);
        NERFART_HIP(e1); NERFART_HIP(e2);
    }
    if (int rc = nerfart_linspace_depths(w.t_coarse, n_samples, nullptr, nullptr, near_s, far_s, n_rays, w.d_coarse, n_samples, stream)) return rc;
    if (int rc = nerfart_sort_concat(n_rays, w.d_coarse, n_samples, n_samples, w.d_fine, n_importance, n_importance, d_all, P, stream)) return rc;
    for (int c0 = 0; c0 < n_rays; c0 += k3_rays_chunk) {
        const int rk = (n_rays - c0 < k3_rays_chunk) ? n_rays - c0 : k3_rays_chunk;
        const size_t po = (size_t)c0 * P;
        if (int rc = nerfart_sdf_nabla_fwd_rays(surf_blob, precision, rays_o + 3 * (size_t)c0, w.rays_dn + 3 * (size_t)c0, nullptr,
                                                d_all + po, rk, P, P, R_bg, sdf + po, nabla + 3 * po, w.h7, w.nabla_ws, (long long)w.nabla_ws_bytes, stream)) return rc;
        if (int rc = nerfart_radiance_fwd_rays(rad_blob, rad_precision, view_tiles, rays_o + 3 * (size_t)c0, w.rays_dn + 3 * (size_t)c0,
                                               nullptr, d_all + po, rk, P, P, nabla + 3 * po, w.h7, rad + 3 * po, stream)) return rc;
    }
    return nerfart_volsdf_composite(n_rays, P, d_all, sdf, rad, nabla, alpha, beta, white_bkgd, rgb, depth, acc, normals,
                                    sigma_out, p_out, tau_out, stream);
}

int nerfart_volsdf_render_staged_fwd(const float* surf_blob, int precision, const float* rad_blob, int rad_precision, const float* sampler_blob,
                                     int sampler_precision, float sampler_guard, int view_tiles, const float* rays_o,
                              const float* rays_d, int n_rays, float near_s, float far_s, float R_bg, float alpha,
                              float beta, float eps, int n_samples, int n_importance, int max_upsample_steps,
                              int max_bisection_steps, int white_bkgd, int k3_rays_chunk, const float* t_coarse_dev,
                              const float* t_init_dev, const float* u_up_dev, const float* u_final_dev, int u_final_per_ray,
                              float* rgb, float* depth, float* acc, float* normals, float* d_all_out, float* sdf_out, float* nabla_out,
                              float* radiance_out, float* sigma_out, float* p_out, float* tau_out, float* beta_map_out,
                              float* iter_usage_out, int* n_escalated, void* workspace, long long workspace_bytes, void* stream_) {
    return nerfart_volsdf_render_staged2_fwd(surf_blob, precision, rad_blob, rad_precision, sampler_blob, sampler_precision, sampler_guard, 0, view_tiles, rays_o,
                                             rays_d, n_rays, near_s, far_s, R_bg, alpha, beta, eps, n_samples, n_importance, max_upsample_steps,
                                             max_bisection_steps, white_bkgd, k3_rays_chunk, t_coarse_dev, t_init_dev, u_up_dev, u_final_dev, u_final_per_ray,
                                             rgb, depth, acc, normals, d_all_out, sdf_out, nabla_out, radiance_out, sigma_out, p_out, tau_out, beta_map_out,
                                             iter_usage_out, n_escalated, workspace, workspace_bytes, stream_);
}

int nerfart_volsdf_render_mixed_fwd(const float* surf_blob, const float* rad_blob, int precision, const float* sampler_blob, int sampler_precision,
                                    int view_tiles, const float* rays_o,
                              const float* rays_d, int n_rays, float near_s, float far_s, float R_bg, float alpha,
                              float beta, float eps, int n_samples, int n_importance, int max_upsample_steps,
                              int max_bisection_steps, int white_bkgd, int k3_rays_chunk, const float* t_coarse_dev,
                              const float* t_init_dev, const float* u_up_dev, const float* u_final_dev, int u_final_per_ray,
                              float* rgb, float* depth, float* acc, float* normals, float* d_all_out, float* sdf_out, float* nabla_out,
                              float* radiance_out, float* sigma_out, float* p_out, float* tau_out, float* beta_map_out,
                              float* iter_usage_out, void* workspace, long long workspace_bytes, void* stream_) {
    return nerfart_volsdf_render_staged_fwd(surf_blob, precision, rad_blob, precision, sampler_blob, sampler_precision, 0.f, view_tiles, rays_o, rays_d, n_rays,
                                            near_s, far_s, R_bg, alpha, beta, eps, n_samples, n_importance, max_upsample_steps, max_bisection_steps, white_bkgd,
                                            k3_rays_chunk, t_coarse_dev, t_init_dev, u_up_dev, u_final_dev, u_final_per_ray, rgb, depth, acc, normals, d_all_out,
                                            sdf_out, nabla_out, radiance_out, sigma_out, p_out, tau_out, beta_map_out, iter_usage_out, nullptr, workspace,
                                            workspace_bytes, stream_);
}

int nerfart_volsdf_render_fwd(const float* surf_blob, const float* rad_blob, int precision, int view_tiles, const float* rays_o,
                              const float* rays_d, int n_rays, float near_s, float far_s, float R_bg, float alpha,
                              float beta, float eps, int n_samples, int n_importance, int max_upsample_steps,
                              int max_bisection_steps, int white_bkgd, int k3_rays_chunk, const float* t_coarse_dev,
                              const float* t_init_dev, const float* u_up_dev, const float* u_final_dev, int u_final_per_ray,
                              float* rgb, float* depth, float* acc, float* normals, float* d_all_out, float* sdf_out, float* nabla_out,
                              float* radiance_out, float* sigma_out, float* p_out, float* tau_out, float* beta_map_out,
                              float* iter_usage_out, void* workspace, long long workspace_bytes, void* stream_) {
    return nerfart_volsdf_render_mixed_fwd(surf_blob, rad_blob, precision, surf_blob, precision, view_tiles, rays_o, rays_d, n_rays, near_s, far_s, R_bg, alpha,
                                           beta, eps, n_samples, n_importance, max_upsample_steps, max_bisection_steps, white_bkgd, k3_rays_chunk,
                                           t_coarse_dev, t_init_dev, u_up_dev, u_final_dev, u_final_per_ray, rgb, depth, acc, normals, d_all_out, sdf_out,
                                           nabla_out, radiance_out, sigma_out, p_out, tau_out, beta_map_out, iter_usage_out, workspace, workspace_bytes,
                                           stream_);
}

}  // extern "C"
