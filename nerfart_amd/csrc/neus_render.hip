// neus_render.hip - NeuS sampler ('official_solution' / 'direct_use' / 'direct_more' up-sampling), compositor and render
// orchestrator (reference models/frameworks/neus.py: cdf_Phi_s/sdf_to_alpha/alpha_to_w :29-78,
// volume_render :142-424 with the up-sampling loop :275-303; utils/rend_util.py
// near_far_from_sphere :168-186, sample_pdf :256-293).
//
// One 64-lane wave (= one workgroup) per ray, samples in LDS, as in volsdf_render.hip.
#include "ray_common.h"
#include <stdlib.h>

namespace nerfart {

// near = max(-o.d - r, 0), far = max(-o.d + r, r)   (rend_util.py:176-184)
__global__ void k_near_far(const float* __restrict__ o, const float* __restrict__ d, int n, float r,
                           float* __restrict__ near, float* __restrict__ far) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float dot = o[3 * i] * d[3 * i] + o[3 * i + 1] * d[3 * i + 1] + o[3 * i + 2] * d[3 * i + 2];
    const float mid = -dot;
    near[i] = fmaxf(mid - r, 0.f);
    far[i] = fmaxf(mid + r, r);
}

__global__ void k_midpoints(const float* __restrict__ d, int P, int n_rays, float* __restrict__ mid) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n_rays * (P - 1)) return;
    const int r = (int)(i / (P - 1)), k = (int)(i - (long long)r * (P - 1));
    mid[i] = 0.5f * (d[(size_t)r * P + k + 1] + d[(size_t)r * P + k]);
}

// One up-sampling round (neus.py:279-296): slope-limited SDF estimate at the interval ends,
// sigmoid CDF with fixed inv_s, alpha -> visibility weights, n_new inverse-CDF samples.
// DIRECT ('direct_use' / 'direct_more', neus.py:242-269): the weights are sdf_to_w(sdf, s) of the bins themselves (:36-63): the
// opacity of consecutive sigmoid CDFs, clamped at 0, no slope estimate.
template <bool DIRECT>
__global__ void __launch_bounds__(64)
k_neus_upsample(int n, int cap, int n_new, float inv_s, const float* __restrict__ dA, const float* __restrict__ sA,
                const float* __restrict__ u_new, int u_new_stride, float* __restrict__ d_new) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ray = blockIdx.x, lane = threadIdx.x;
    u_new += (size_t)ray * u_new_stride;            // 0: the shared linspace table (det); else this ray's random numbers
    float* d = sm; float* s = sm + n; float* w = sm + 2 * n; float* cdf = sm + 3 * n; float* out = sm + 4 * n;
    for (int i = lane; i < n; i += 64) { d[i] = dA[(size_t)ray * cap + i]; s[i] = sA[(size_t)ray * cap + i]; }
    __syncthreads();
    const int nint = n - 1, seg = (nint + 63) >> 6, k0 = lane * seg, k1 = (k0 + seg < nint) ? k0 + seg : nint;
    // alpha per interval -> w[k] (temporarily), local product of (1 - alpha + 1e-10)
    float lp = 1.f;
    for (int k = k0; k < k1; ++k) {
        float a;
        if (DIRECT) {
            const float pc = sigmoidf_(s[k] * inv_s), nc = sigmoidf_(s[k + 1] * inv_s);
            a = fmaxf((pc - nc) / (pc + 1e-10f), 0.f);
        } else {
            const float ps = s[k], ns = s[k + 1], pz = d[k], nz = d[k + 1];
            const float mid = (ps + ns) * 0.5f;
            const float dot = (ns - ps) / (nz - pz + 1e-5f);
            float pdot = 0.f;
            if (k > 0) pdot = (s[k] - s[k - 1]) / (d[k] - d[k - 1] + 1e-5f);
            const float dv = fminf(fmaxf(fminf(pdot, dot), -10.f), 0.f);
            const float dist = nz - pz;
            const float pe = mid - dv * dist * 0.5f, ne = mid + dv * dist * 0.5f;
            const float pc = sigmoidf_(pe * inv_s), nc = sigmoidf_(ne * inv_s);
            a = (pc - nc + 1e-5f) / (pc + 1e-5f);
        }
        w[k] = a;
        lp *= (1.f - a + 1e-10f);
    }
    float T = wave_excl_prod(lp);
    float part = 0.f;
    for (int k = k0; k < k1; ++k) {
        const float a = w[k];
        const float wk = a * T + 1e-5f;          // alpha_to_w, then + 1e-5 of sample_pdf
        T *= (1.f - a + 1e-10f);
        w[k] = wk;
        part += wk;
    }
    const float total = wave_sum(part);
    float ps = 0.f;
    for (int k = k0; k < k1; ++k) ps += w[k] / total;
    float run = wave_excl_sum(ps);
    if (lane == 0) cdf[0] = 0.f;
    for (int k = k0; k < k1; ++k) { run += w[k] / total; cdf[k + 1] = run; }
    __syncthreads();
    const int npad = n_new < 64 ? 64 : n_new;        // bitonic_sort needs >= 1 element per step; pad with +inf
    for (int j = lane; j < npad; j += 64) out[j] = (j < n_new) ? invert_cdf_at(d, cdf, n, u_new[j]) : INFINITY;
    bitonic_sort(out, npad);
    for (int j = lane; j < n_new; j += 64) d_new[(size_t)ray * n_new + j] = out[j];
}

// In-place stable merge of n_new sorted (depth, sdf) pairs into a ray's n sorted pairs (neus.py:297-302).
__global__ void __launch_bounds__(64)
k_merge_pairs(int n, int cap, int n_new, float* __restrict__ dA, float* __restrict__ sA,
              const float* __restrict__ d_new, const float* __restrict__ s_new) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ray = blockIdx.x;
    float* d_old = sm; float* s_old = sm + n; float* dn = sm + 2 * n; float* sn = dn + n_new;
    for (int i = threadIdx.x; i < n; i += 64) { d_old[i] = dA[(size_t)ray * cap + i]; s_old[i] = sA[(size_t)ray * cap + i]; }
    for (int i = threadIdx.x; i < n_new; i += 64) { dn[i] = d_new[(size_t)ray * n_new + i]; sn[i] = s_new[(size_t)ray * n_new + i]; }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 64) {
        const int r = i + lower_bound(dn, n_new, d_old[i]);
        dA[(size_t)ray * cap + r] = d_old[i]; sA[(size_t)ray * cap + r] = s_old[i];
    }
    for (int k = threadIdx.x; k < n_new; k += 64) {
        const int r = k + upper_bound(d_old, n, dn[k]);
        dA[(size_t)ray * cap + r] = dn[k]; sA[(size_t)ray * cap + r] = sn[k];
    }
}

// NeuS ray integration (neus.py:322, :373-395): cdf = sigmoid(sdf * s), alpha from consecutive cdfs,
// w = alpha * cumprod(1 - alpha + 1e-10), radiance sampled at the interval mid-points.
__global__ void __launch_bounds__(64)
k_composite_neus(int P, const float* __restrict__ d_all, const float* __restrict__ sdf,
                 const float* __restrict__ rad_mid, const float* __restrict__ nabla, float s_inv, int white_bkgd,
                 float* __restrict__ rgb, float* __restrict__ depth, float* __restrict__ acc,
                 float* __restrict__ normals, float* __restrict__ cdf_out, float* __restrict__ alpha_out,
                 float* __restrict__ w_out, float* __restrict__ dmid_out) {
    const int ray = blockIdx.x, lane = threadIdx.x;
    const int nint = P - 1, seg = (nint + 63) >> 6, k0 = lane * seg, k1 = (k0 + seg < nint) ? k0 + seg : nint;
    const float* dr = d_all + (size_t)ray * P;
    const float* sr = sdf + (size_t)ray * P;
    float lp = 1.f;
    for (int k = k0; k < k1; ++k) {
        const float c0 = sigmoidf_(sr[k] * s_inv), c1 = sigmoidf_(sr[k + 1] * s_inv);
        const float a = fmaxf((c0 - c1) / (c0 + 1e-10f), 0.f);
        lp *= (1.f - a + 1e-10f);
    }
    float T = wave_excl_prod(lp);
    const float T0 = T;
    float r = 0.f, g = 0.f, b = 0.f, asum = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
    for (int k = k0; k < k1; ++k) {
        const float c0 = sigmoidf_(sr[k] * s_inv), c1 = sigmoidf_(sr[k + 1] * s_inv);
        const float a = fmaxf((c0 - c1) / (c0 + 1e-10f), 0.f);
        const float wk = a * T;
        T *= (1.f - a + 1e-10f);
        const size_t q = (size_t)ray * nint + k;
        r += wk * rad_mid[3 * q]; g += wk * rad_mid[3 * q + 1]; b += wk * rad_mid[3 * q + 2];
        asum += wk;
        if (normals) {
            const size_t qp = (size_t)ray * P + k;
            const float vx = nabla[3 * qp], vy = nabla[3 * qp + 1], vz = nabla[3 * qp + 2];
            const float nr = fmaxf(sqrtf(vx * vx + vy * vy + vz * vz), 1e-12f);
            nx += vx / nr * wk; ny += vy / nr * wk; nz += vz / nr * wk;
        }
        if (cdf_out) cdf_out[(size_t)ray * P + k] = c0;
        if (alpha_out) alpha_out[q] = a;
        if (w_out) w_out[q] = wk;
        if (dmid_out) dmid_out[q] = 0.5f * (dr[k + 1] + dr[k]);
    }
    if (cdf_out && lane == 0) cdf_out[(size_t)ray * P + P - 1] = sigmoidf_(sr[P - 1] * s_inv);
    r = wave_sum(r); g = wave_sum(g); b = wave_sum(b); asum = wave_sum(asum);
    T = T0;
    float dp = 0.f;
    const float inv = asum + 1e-10f;
    for (int k = k0; k < k1; ++k) {
        const float c0 = sigmoidf_(sr[k] * s_inv), c1 = sigmoidf_(sr[k + 1] * s_inv);
        const float a = fmaxf((c0 - c1) / (c0 + 1e-10f), 0.f);
        const float wk = a * T;
        T *= (1.f - a + 1e-10f);
        dp += wk / inv * (0.5f * (dr[k + 1] + dr[k]));
    }
    dp = wave_sum(dp);
    if (normals) { nx = wave_sum(nx); ny = wave_sum(ny); nz = wave_sum(nz); }
    if (lane == 0) {
        if (white_bkgd) { r += 1.f - asum; g += 1.f - asum; b += 1.f - asum; }
        rgb[3 * (size_t)ray] = r; rgb[3 * (size_t)ray + 1] = g; rgb[3 * (size_t)ray + 2] = b;
        depth[ray] = dp; acc[ray] = asum;
        if (normals) { normals[3 * (size_t)ray] = nx; normals[3 * (size_t)ray + 1] = ny; normals[3 * (size_t)ray + 2] = nz; }
    }
}

static inline size_t align_up_n(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace nerfart

using namespace nerfart;

extern "C" {

int nerfart_sdf_fwd_rays(const float*, int, const float*, const float*, const int*, const float*, int, int, int, float, float*, int, void*);
int nerfart_sdf_nabla_fwd_rays(const float*, int, const float*, const float*, const int*, const float*, int, int, int, float, float*, float*, float*, void*,
                               long long, void*);
long long nerfart_sdf_nabla_workspace_bytes(int precision);
int nerfart_radiance_fwd_rays(const float*, int, int, const float*, const float*, const int*, const float*, int, int, int, const float*, const float*, float*, void*);
int nerfart_normalize_dirs(const float*, float*, int, void*);
int nerfart_linspace_depths(const float*, int, const float*, const float*, float, float, int, float*, int, void*);
void nerfart_linspace(float, float, int, float*);

static int set_lds_n(const void* k, size_t bytes) {
    if (bytes > 160 * 1024) { set_last_error("per-ray kernel needs more than 160 KiB of LDS"); return 2; }
    NERFART_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

int nerfart_near_far_from_sphere(const float* rays_o, const float* rays_dn, int n_rays, float r, float* near,
                                 float* far, void* stream) {
    if (n_rays <= 0) return 0;
    hipLaunchKernelGGL(k_near_far, dim3((n_rays + 255) / 256), dim3(256), 0, (hipStream_t)stream, rays_o, rays_dn, n_rays, r, near, far);
    NERFART_HIP(hipGetLastError());
    return 0;
}

static int upsample_step(bool direct, int n_rays, int n, int cap, int n_new, float inv_s, const float* d, const float* sdf,
                         const float* u_new, int u_new_stride, float* d_new, void* stream) {
    if (n_rays <= 0) return 0;
    if (n < 2 || n_new < 1) { set_last_error("neus upsample: needs n >= 2 bins and n_new >= 1"); return 2; }
    if (n_new > 64 && (n_new & (n_new - 1))) { set_last_error("neus upsample: n_new must be <= 64 or a power of two"); return 2; }
    const int npad = n_new < 64 ? 64 : n_new;
    const size_t lds = ((size_t)4 * n + npad) * sizeof(float);
    const void* fn = direct ? (const void*)k_neus_upsample<true> : (const void*)k_neus_upsample<false>;
    if (int rc = set_lds_n(fn, lds)) return rc;
    if (direct) hipLaunchKernelGGL(k_neus_upsample<true>, dim3(n_rays), dim3(64), lds, (hipStream_t)stream, n, cap, n_new, inv_s, d, sdf, u_new, u_new_stride, d_new);
    else hipLaunchKernelGGL(k_neus_upsample<false>, dim3(n_rays), dim3(64), lds, (hipStream_t)stream, n, cap, n_new, inv_s, d, sdf, u_new, u_new_stride, d_new);
    NERFART_HIP(hipGetLastError());
    return 0;
}

int nerfart_neus_upsample_step(int n_rays, int n, int cap, int n_new, float inv_s, const float* d, const float* sdf,
                               const float* u_new, int u_new_stride, float* d_new, void* stream) {
    return upsample_step(false, n_rays, n, cap, n_new, inv_s, d, sdf, u_new, u_new_stride, d_new, stream);
}

int nerfart_neus_direct_upsample_step(int n_rays, int n, int cap, int n_new, float inv_s, const float* d, const float* sdf,
                                      const float* u_new, int u_new_stride, float* d_new, void* stream) {
    return upsample_step(true, n_rays, n, cap, n_new, inv_s, d, sdf, u_new, u_new_stride, d_new, stream);
}

int nerfart_merge_sorted_pairs(int n_rays, int n, int cap, int n_new, float* d, float* sdf, const float* d_new,
                               const float* s_new, void* stream) {
    if (n_rays <= 0) return 0;
    if (n + n_new > cap) { set_last_error("merge_sorted_pairs: n + n_new exceeds the row capacity"); return 2; }
    const size_t lds = ((size_t)2 * n + 2 * n_new) * sizeof(float);
    if (int rc = set_lds_n((const void*)k_merge_pairs, lds)) return rc;
    hipLaunchKernelGGL(k_merge_pairs, dim3(n_rays), dim3(64), lds, (hipStream_t)stream, n, cap, n_new, d, sdf, d_new, s_new);
    NERFART_HIP(hipGetLastError());
    return 0;
}

int nerfart_neus_composite(int n_rays, int P, const float* d_all, const float* sdf, const float* radiance_mid,
                           const float* nabla, float s, int white_bkgd, float* rgb, float* depth, float* acc,
                           float* normals, float* cdf_out, float* alpha_out, float* w_out, float* d_mid_out,
                           void* stream) {
    if (n_rays <= 0) return 0;
    if (normals && !nabla) { set_last_error("composite: normals requested without nablas"); return 2; }
    hipLaunchKernelGGL(k_composite_neus, dim3(n_rays), dim3(64), 0, (hipStream_t)stream, P, d_all, sdf, radiance_mid, nabla, s,
                       white_bkgd, rgb, depth, acc, normals, cdf_out, alpha_out, w_out, d_mid_out);
    NERFART_HIP(hipGetLastError());
    return 0;
}

typedef struct {
    float *rays_dn, *near, *far, *t_coarse, *u_new, *d, *s, *d_new, *s_new, *d_mid, *sdf, *nabla, *nabla_mid, *sdf_mid, *rad, *h7;
    float *t_more, *d_more, *s_more;   // 'direct_more': the N_nograd_samples table, depths and sdf of the no-gradient samples
    char* nabla_ws;           // softplus' scratch of the reverse-mode grad(SDF) kernel
    size_t nabla_ws_bytes;
} neus_ws_t;

static size_t carve_neus(char* base, int R, int n_samples, int n_imp, int k3_rays, neus_ws_t* w, int n_more = 0) {
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += align_up_n(bytes); return base ? base + r : (char*)nullptr; };
    const int P = n_samples + n_imp;
    float* p;
    p = (float*)take((size_t)R * 12); if (w) w->rays_dn = p;
    p = (float*)take((size_t)R * 4); if (w) w->near = p;
    p = (float*)take((size_t)R * 4); if (w) w->far = p;
    p = (float*)take((size_t)n_samples * 4); if (w) w->t_coarse = p;
    p = (float*)take((size_t)(n_imp + 64) * 4); if (w) w->u_new = p;
    p = (float*)take((size_t)R * P * 4); if (w) w->d = p;
    p = (float*)take((size_t)R * P * 4); if (w) w->s = p;
    p = (float*)take((size_t)R * n_imp * 4); if (w) w->d_new = p;
    p = (float*)take((size_t)R * n_imp * 4); if (w) w->s_new = p;
    p = (float*)take((size_t)R * (P - 1) * 4); if (w) w->d_mid = p;
    p = (float*)take((size_t)R * P * 4); if (w) w->sdf = p;
    p = (float*)take((size_t)R * P * 12); if (w) w->nabla = p;
    const int rk = k3_rays < R ? k3_rays : R;
    p = (float*)take((size_t)rk * (P - 1) * 12); if (w) w->nabla_mid = p;
    p = (float*)take((size_t)rk * (P - 1) * 4); if (w) w->sdf_mid = p;
    p = (float*)take((size_t)R * (P - 1) * 12); if (w) w->rad = p;
    p = (float*)take((size_t)rk * (P - 1) * 256 * 4); if (w) w->h7 = p;
    const long long nb0 = nerfart_sdf_nabla_workspace_bytes(0), nb1 = nerfart_sdf_nabla_workspace_bytes(1);
    const size_t nb = (size_t)(nb0 > nb1 ? nb0 : nb1);
    char* np = take(nb);
    if (w) { w->nabla_ws = np; w->nabla_ws_bytes = nb; }
    // (appended: the layout of everything above does not depend on the up-sampling algorithm)
    p = (float*)take((size_t)n_more * 4); if (w) w->t_more = p;
    p = (float*)take((size_t)R * n_more * 4); if (w) w->d_more = p;
    p = (float*)take((size_t)R * n_more * 4); if (w) w->s_more = p;
    return o;
}

long long nerfart_neus_render_workspace_bytes(int n_rays, int n_samples, int n_importance, int k3_rays_chunk) {
    return (long long)carve_neus(nullptr, n_rays, n_samples, n_importance, k3_rays_chunk, nullptr);
}

long long nerfart_neus_render_algo_workspace_bytes(int n_rays, int n_samples, int n_importance, int k3_rays_chunk, int upsample_algo,
                                                   int n_nograd_samples) {
    return (long long)carve_neus(nullptr, n_rays, n_samples, n_importance, k3_rays_chunk, nullptr, upsample_algo == 2 ? n_nograd_samples : 0);
}

// NeuS volume_render for one chunk of rays (N_outside = 0).  upsample_algo: 0 'official_solution' (neus.py:275-303: n_upsample_iters
// rounds of n_importance / n_upsample_iters slope-estimated samples at s = 64 * 2^i), 1 'direct_use' (:242-255: all n_importance samples at
// once from the coarse samples' own visibility weights at s = 1 / fixed_s_recp), 2 'direct_more' (:259-269: the same from n_nograd_samples
// evenly spaced no-gradient samples).  u_new_dev: the uniform numbers - a shared table of n_new = n_importance / n_upsample_iters (algo 0)
// or n_importance (algo 1, 2) values when u_new_per_ray == 0, else [n_rays, n_importance].  t_coarse_dev / t_nograd_dev: torch.linspace(0, 1,
// n_samples / n_nograd_samples) on the device (nerfart_linspace) or NULL (built here, one stream synchronisation).
// s = exp(ln_s * speed_factor).  Outputs as nerfart_volsdf_render_fwd; detailed outputs: d_all/sdf/cdf [R,P],
// nabla [R,P,3], radiance/alpha/w/d_mid on the P-1 mid-points.
int nerfart_neus_render_algo_fwd(const float* surf_blob, const float* rad_blob, int precision, int view_tiles, const float* rays_o,
                                 const float* rays_d, int n_rays, float obj_bounding_radius, float s, int n_samples,
                                 int n_importance, int n_upsample_iters, int upsample_algo, int n_nograd_samples, float fixed_s_recp,
                                 int white_bkgd, int k3_rays_chunk,
                                 const float* t_coarse_dev, const float* t_nograd_dev, const float* u_new_dev, int u_new_per_ray, float* rgb,
                                 float* depth, float* acc, float* normals, float* d_all_out, float* sdf_out,
                                 float* nabla_out, float* radiance_out, float* cdf_out, float* alpha_out, float* w_out,
                                 float* d_mid_out, void* workspace, long long workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rays <= 0) return 0;
    if (upsample_algo < 0 || upsample_algo > 2) { set_last_error("neus render: upsample_algo must be 0 (official_solution), 1 (direct_use) or 2 (direct_more)"); return 2; }
    if (upsample_algo != 0) n_upsample_iters = 1;                 // all n_importance samples in one inversion
    if (n_samples < 2 || n_upsample_iters < 1 || n_importance % n_upsample_iters || k3_rays_chunk < 1) {
        set_last_error("neus render: bad sample counts"); return 2;
    }
    if (upsample_algo == 2 && n_nograd_samples < 2) { set_last_error("neus render: direct_more needs n_nograd_samples >= 2"); return 2; }
    // direct_more holds a ray's n_nograd_samples bins in LDS (4 rows + the new depths): refuse up front, with the number that fits, what would only fail
    // as a generic LDS error from inside the render (the reference takes any N_nograd_samples, the shipped configs use 2048)
    if (upsample_algo == 2 && ((size_t)4 * n_nograd_samples + (n_importance < 64 ? 64 : n_importance)) * sizeof(float) > 160 * 1024) {
        set_last_error("neus render: upsample_algo 'direct_more' keeps 4 x N_nograd_samples floats per ray in the 160 KiB LDS: N_nograd_samples must be <= 10,200 "
                       "(the reference configs use 2048)");
        return 2;
    }
    if (upsample_algo != 0 && !(fixed_s_recp > 0.f)) { set_last_error("neus render: fixed_s_recp must be positive"); return 2; }
    const int P = n_samples + n_importance, n_new = n_importance / n_upsample_iters;
    const int n_more = upsample_algo == 2 ? n_nograd_samples : 0;
    if (u_new_per_ray && !u_new_dev) { set_last_error("neus render: u_new_per_ray needs u_new_dev [n_rays, n_importance]"); return 2; }
    neus_ws_t w;
    const size_t need = carve_neus((char*)workspace, n_rays, n_samples, n_importance, k3_rays_chunk, &w, n_more);
    if (!workspace || (size_t)workspace_bytes < need) { set_last_error("neus render: workspace too small"); return 2; }
    float* sdf = sdf_out ? sdf_out : w.sdf;
    float* nabla = nabla_out ? nabla_out : w.nabla;
    float* rad = radiance_out ? radiance_out : w.rad;
    if (t_coarse_dev && u_new_dev) {
        w.t_coarse = const_cast<float*>(t_coarse_dev); w.u_new = const_cast<float*>(u_new_dev);
    } else {
        float* h = (float*)malloc(sizeof(float) * (size_t)(n_samples + n_new));
        if (!h) { set_last_error("out of host memory"); return 3; }
        nerfart_linspace(0.f, 1.f, n_samples, h);
        nerfart_linspace(0.f, 1.f, n_new, h + n_samples);
        hipError_t e1 = hipMemcpyAsync(w.t_coarse, h, sizeof(float) * n_samples, hipMemcpyHostToDevice, stream);
        hipError_t e2 = hipMemcpyAsync(w.u_new, h + n_samples, sizeof(float) * n_new, hipMemcpyHostToDevice, stream);
        hipError_t e3 = hipStreamSynchronize(stream);
        free(h);
        NERFART_HIP(e1); NERFART_HIP(e2); NERFART_HIP(e3);
        if (u_new_per_ray) w.u_new = const_cast<float*>(u_new_dev);
    }
    if (int rc = nerfart_normalize_dirs(rays_d, w.rays_dn, n_rays, stream)) return rc;
    if (int rc = nerfart_near_far_from_sphere(rays_o, w.rays_dn, n_rays, obj_bounding_radius, w.near, w.far, stream)) return rc;
    if (int rc = nerfart_linspace_depths(w.t_coarse, n_samples, w.near, w.far, 0.f, 0.f, n_rays, w.d, P, stream)) return rc;
    if (int rc = nerfart_sdf_fwd_rays(surf_blob, precision, rays_o, w.rays_dn, nullptr, w.d, n_rays, n_samples, P, 0.f, w.s, P, stream)) return rc;
    int n = n_samples;
    if (upsample_algo == 0) {
        for (int i = 0; i < n_upsample_iters; ++i) {
            if (int rc = nerfart_neus_upsample_step(n_rays, n, P, n_new, 64.f * (float)(1 << i), w.d, w.s,
                                                    u_new_per_ray ? w.u_new + (size_t)i * n_new : w.u_new, u_new_per_ray ? n_importance : 0,
                                                    w.d_new, stream)) return rc;
            if (int rc = nerfart_sdf_fwd_rays(surf_blob, precision, rays_o, w.rays_dn, nullptr, w.d_new, n_rays, n_new, n_new, 0.f, w.s_new, n_new, stream)) return rc;
            if (int rc = nerfart_merge_sorted_pairs(n_rays, n, P, n_new, w.d, w.s, w.d_new, w.s_new, stream)) return rc;
            n += n_new;
        }
    } else {
        const float* bins_d = w.d; const float* bins_s = w.s; int n_bins = n_samples, bins_cap = P;
        if (upsample_algo == 2) {                                 // neus.py:260-263: N_nograd_samples evenly spaced samples, sdf without gradient
            if (t_nograd_dev) {
                w.t_more = const_cast<float*>(t_nograd_dev);
            } else {
                float* h = (float*)malloc(sizeof(float) * (size_t)n_more);
                if (!h) { set_last_error("out of host memory"); return 3; }
                nerfart_linspace(0.f, 1.f, n_more, h);
                hipError_t e1 = hipMemcpyAsync(w.t_more, h, sizeof(float) * n_more, hipMemcpyHostToDevice, stream);
                hipError_t e2 = hipStreamSynchronize(stream);
                free(h);
                NERFART_HIP(e1); NERFART_HIP(e2);
            }
            if (int rc = nerfart_linspace_depths(w.t_more, n_more, w.near, w.far, 0.f, 0.f, n_rays, w.d_more, n_more, stream)) return rc;
            if (int rc = nerfart_sdf_fwd_rays(surf_blob, precision, rays_o, w.rays_dn, nullptr, w.d_more, n_rays, n_more, n_more, 0.f, w.s_more, n_more, stream)) return rc;
            bins_d = w.d_more; bins_s = w.s_more; n_bins = n_more; bins_cap = n_more;
        }
        if (int rc = nerfart_neus_direct_upsample_step(n_rays, n_bins, bins_cap, n_new, 1.f / fixed_s_recp, bins_d, bins_s, w.u_new,
                                                       u_new_per_ray ? n_importance : 0, w.d_new, stream)) return rc;
        // d_all = sort(cat(d_coarse, d_fine)) (:254-255, :268-269); the fine samples' sdf rides along so that w.s stays the sdf row of w.d
        if (int rc = nerfart_sdf_fwd_rays(surf_blob, precision, rays_o, w.rays_dn, nullptr, w.d_new, n_rays, n_new, n_new, 0.f, w.s_new, n_new, stream)) return rc;
        if (int rc = nerfart_merge_sorted_pairs(n_rays, n, P, n_new, w.d, w.s, w.d_new, w.s_new, stream)) return rc;
        n += n_new;
    }
    {
        const long long tot = (long long)n_rays * (P - 1);
        hipLaunchKernelGGL(k_midpoints, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, w.d, P, n_rays, w.d_mid);
        NERFART_HIP(hipGetLastError());
    }
    if (d_all_out) NERFART_HIP(hipMemcpyAsync(d_all_out, w.d, sizeof(float) * (size_t)n_rays * P, hipMemcpyDeviceToDevice, stream));
    // sdf + nablas at the P sample points (neus.py:320).  The nablas feed only normals_volume and the detailed output (:385-392):
    // when neither is asked for, the sampler's own sdf row IS the result - the same network at the same sorted depths, evaluated by
    // K2 during the up-sampling rounds (the reverse-mode kernel's forward sweep is K2's arithmetic) - and no launch is needed
    // (only at precision 1, where the two kernels' forward sweeps are the same code and tests hold the pixels bit-identical; elsewhere the
    // samples' sdf always comes from the SDF + nabla kernel, so pixels never depend on calc_normal)
    if (normals || nabla_out || precision != 1) {
        if (int rc = nerfart_sdf_nabla_fwd_rays(surf_blob, precision, rays_o, w.rays_dn, nullptr, w.d, n_rays, P, P, 0.f, sdf, nabla, nullptr, w.nabla_ws, (long long)w.nabla_ws_bytes, stream)) return rc;
    } else {
        NERFART_HIP(hipMemcpyAsync(sdf, w.s, sizeof(float) * (size_t)n_rays * P, hipMemcpyDeviceToDevice, stream));
    }
    // radiance at the P-1 mid-points, with their own nablas (neus.py:324 -> forward_radiance :111-114)
    for (int c0 = 0; c0 < n_rays; c0 += k3_rays_chunk) {
        const int rk = (n_rays - c0 < k3_rays_chunk) ? n_rays - c0 : k3_rays_chunk;
        const size_t po = (size_t)c0 * (P - 1);
        if (int rc = nerfart_sdf_nabla_fwd_rays(surf_blob, precision, rays_o + 3 * (size_t)c0, w.rays_dn + 3 * (size_t)c0, nullptr,
                                                w.d_mid + po, rk, P - 1, P - 1, 0.f, w.sdf_mid, w.nabla_mid, w.h7, w.nabla_ws, (long long)w.nabla_ws_bytes, stream)) return rc;
        if (int rc = nerfart_radiance_fwd_rays(rad_blob, precision, view_tiles, rays_o + 3 * (size_t)c0, w.rays_dn + 3 * (size_t)c0, nullptr,
                                               w.d_mid + po, rk, P - 1, P - 1, w.nabla_mid, w.h7, rad + 3 * po, stream)) return rc;
    }
    return nerfart_neus_composite(n_rays, P, w.d, sdf, rad, nabla, s, white_bkgd, rgb, depth, acc, normals, cdf_out,
                                  alpha_out, w_out, d_mid_out, stream);
}

// upsample_algo = 'official_solution' (what the four shipped configs use): the entry point of ABI versions 1 - 2.
int nerfart_neus_render_fwd(const float* surf_blob, const float* rad_blob, int precision, int view_tiles, const float* rays_o,
                            const float* rays_d, int n_rays, float obj_bounding_radius, float s, int n_samples,
                            int n_importance, int n_upsample_iters, int white_bkgd, int k3_rays_chunk,
                            const float* t_coarse_dev, const float* u_new_dev, int u_new_per_ray, float* rgb,
                            float* depth, float* acc, float* normals, float* d_all_out, float* sdf_out,
                            float* nabla_out, float* radiance_out, float* cdf_out, float* alpha_out, float* w_out,
                            float* d_mid_out, void* workspace, long long workspace_bytes, void* stream_) {
    return nerfart_neus_render_algo_fwd(surf_blob, rad_blob, precision, view_tiles, rays_o, rays_d, n_rays, obj_bounding_radius, s, n_samples,
                                        n_importance, n_upsample_iters, 0, 0, 1.f / 64.f, white_bkgd, k3_rays_chunk, t_coarse_dev, nullptr, u_new_dev,
                                        u_new_per_ray, rgb, depth, acc, normals, d_all_out, sdf_out, nabla_out, radiance_out, cdf_out, alpha_out,
                                        w_out, d_mid_out, workspace, workspace_bytes, stream_);
}

}  // extern "C"
