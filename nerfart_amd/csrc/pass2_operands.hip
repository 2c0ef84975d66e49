// pass2_operands.hip - the per-point glue of the fine-tune step's pass 2 (row a19; reference: what autograd does between the
// network calls of models/frameworks/volsdf.py:759-770 / neus.py:520-576) as a handful of single-pass kernels:
//
//   k_ray_points          pts = o + d t, view = d per sample                                   (volsdf.py:503-506)
//   k_volsdf_cotangents   the cotangents the second-order SDF sweep starts from: sbar (the compositor's d loss / d sdf, zero where
//                         the sphere clamp sdf = min(net, R - |x|) took the other branch, volsdf.py:97-100), nbar = the radiance net's
//                         d loss / d normal + the eikonal term's w 2 (|n| - 1) / N_patch n / |n| (volsdf.py:764-768), and the
//                         eikonal loss per ray
//   k_operand_*           the NARROW operands of the weight-gradient reductions (csrc/wgrad.hip, 64 bf16 columns): the encoding of
//                         the points with its tangent along nbar (the input side of SDF layers 0 and 4, models/base.py:46-64), the
//                         radiance net's raw inputs [x | v | n] (its layer 0), its output delta g_rgb rgb (1 - rgb) (the sigmoid,
//                         base.py:281) and [sbar; 1] (the sdf row of the last SDF layer).  hi parts in columns 0 .. c-1 and, when
//                         c <= 32, lo parts x - bf16(x) in columns 32 .. 32+c-1 (the caller adds the two halves of the result).
//
// All HBM bound and tiny next to the dumps (<= 128 B written per row); they replace ~100 ATen launches per launch group (torch.sin /
// cos / stack / cat / zeros / slice copies / casts over [M, 39] tensors; 8,400 launches and ~65 ms per 480x270 step in
// profiles/r04c_train_kernel_stats.txt).  One thread writes 16 bytes (8 columns of one row): a wave covers 8 consecutive rows =
// 1 KiB contiguous.
#include "nerfart_common.h"

namespace nerfart {
namespace p2 {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// round to nearest even, as torch's .to(torch.bfloat16) (finite inputs)
__device__ __forceinline__ unsigned bf16_bits(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_value(unsigned bits) { return __uint_as_float(bits << 16); }

// column k of embed(x) (base.py:46-64: [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...]) or, with `tangent`, of d embed(x)/dx . d
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 load3(const float* p, long long m) { return V3{p[3 * m], p[3 * m + 1], p[3 * m + 2]}; }
__device__ __forceinline__ float pick(const V3& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }   // (no indexed private arrays)
__device__ __forceinline__ float embed_col(const V3& x, const V3& d, int k, int multires, bool tangent) {
    if (k < 3) return tangent ? pick(d, k) : pick(x, k);
    const int j = k - 3, oct = j / 6, w = j - 6 * oct, a = w < 3 ? w : w - 3;
    if (multires < 0 || oct >= multires) return 0.f;
    const float f = (float)(1u << oct);
    const float xf = pick(x, a) * f;                         // exact: a power of two
    if (!tangent) return w < 3 ? sinf(xf) : cosf(xf);
    const float df = pick(d, a) * f;
    return w < 3 ? cosf(xf) * df : -sinf(xf) * df;
}
__device__ __forceinline__ int embed_width(int multires) { return multires < 0 ? 3 : 3 + 6 * multires; }

// 8 columns [8 q, 8 q + 8) of a row whose fp32 column k is val(k) for k < c (0 beyond), packed hi | (split) lo at +32
template <class F>
__device__ __forceinline__ u32x4 pack8(int q, int c, bool split, F val) {
    u32x4 out = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int col = 8 * q + e;
        const bool lo_half = split && col >= 32;
        const int k = lo_half ? col - 32 : col;
        unsigned b = 0u;
        if (k < c) {
            const float x = val(k);
            const unsigned hi = bf16_bits(x);
            b = lo_half ? bf16_bits(x - bf16_value(hi)) : hi;
        }
        out[e >> 1] |= b << (16 * (e & 1));
    }
    return out;
}

__global__ void __launch_bounds__(256)
k_ray_points(const float* __restrict__ o, const float* __restrict__ dn, const float* __restrict__ depth, long long R, int P,
             float* __restrict__ pts, float* __restrict__ view) {
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= R * P) return;
    const long long r = m / P;
    const float t = depth[m];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float d = dn[3 * r + a];
        float dt = d * t;
        asm volatile("" : "+v"(dt));                         // no fma: two roundings, as torch's o + d * t (pass 1 kept its state at THOSE points)
        pts[3 * m + a] = o[3 * r + a] + dt;
        if (view) view[3 * m + a] = d;
    }
}

// one wave per ray
__global__ void __launch_bounds__(64)
k_volsdf_cotangents(const float* __restrict__ pts, const float* __restrict__ sdf, const float* __restrict__ g_sdf,
                    const float* __restrict__ nab, const float* __restrict__ g_n, const float* __restrict__ g_n_extra, long long R, int P,
                    float R_bg, float w_eik, long long group_rays, float* __restrict__ sbar, float* __restrict__ nbar,
                    float* __restrict__ eik_ray) {
    const long long r = blockIdx.x;
    const int lane = threadIdx.x;
    // the reference's pass 2 takes the eikonal MEAN over each patch of group_rays rays (the last patch of a launch may be ragged)
    float coef = 0.f;
    if (w_eik != 0.f) {
        const long long g = (group_rays <= 0 || group_rays > R) ? R : group_rays;
        const long long tail = R % g;
        const long long n_in_patch = (tail && r >= R - tail) ? tail : g;
        coef = 1.0f / ((float)n_in_patch * (float)P);
    }
    float acc = 0.f;
    for (int p = lane; p < P; p += 64) {
        const long long m = r * P + p;
        const float x = pts[3 * m], y = pts[3 * m + 1], z = pts[3 * m + 2];
        const float s = sdf[m];
        const bool clamped = R_bg > 0.f && s >= (R_bg - sqrtf(x * x + y * y + z * z)) - 1e-6f;      // R_bg <= 0: no sphere (NeuS)
        sbar[m] = clamped ? 0.f : g_sdf[m];
        const float nx = nab[3 * m], ny = nab[3 * m + 1], nz = nab[3 * m + 2];
        float bx = 0.f, by = 0.f, bz = 0.f;
        if (g_n) { bx = g_n[3 * m]; by = g_n[3 * m + 1]; bz = g_n[3 * m + 2]; }
        if (w_eik != 0.f) {
            const float nn = sqrtf(nx * nx + ny * ny + nz * nz);
            const float err = nn - 1.0f;
            acc = fmaf(err, err, acc);
            const float k = (2.0f * w_eik) * coef * err / nn;
            bx = fmaf(k, nx, bx); by = fmaf(k, ny, by); bz = fmaf(k, nz, bz);
        }
        if (g_n_extra) { bx += g_n_extra[3 * m]; by += g_n_extra[3 * m + 1]; bz += g_n_extra[3 * m + 2]; }
        nbar[3 * m] = bx; nbar[3 * m + 1] = by; nbar[3 * m + 2] = bz;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) eik_ray[r] = w_eik * coef * acc;
}

// rows [0, M): embed(pts); rows [rows_pad, rows_pad + M): its tangent along dir; every other row of the 2 rows_pad zero
__global__ void __launch_bounds__(256)
k_operand_embed_pair(const float* __restrict__ pts, const float* __restrict__ dir, long long M, long long rows_pad, int multires,
                     u32x4* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long row = i >> 3;
    const int q = (int)(i & 7);
    if (row >= 2 * rows_pad) return;
    const bool tangent = row >= rows_pad;
    const long long m = tangent ? row - rows_pad : row;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (m < M) {
        const V3 x = load3(pts, m), d = load3(dir, m);
        const int c = embed_width(multires);
        v = pack8(q, c, c <= 32, [&](int k) { return embed_col(x, d, k, multires, tangent); });
    }
    out[i] = v;
}

// rows [0, M): [embed(x, mx) | embed(v, mv) | n]; rows [M, rows_pad) zero
__global__ void __launch_bounds__(256)
k_operand_inputs(const float* __restrict__ xs, int mx, const float* __restrict__ vs, int mv, const float* __restrict__ ns, long long M,
                 long long rows_pad, u32x4* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long m = i >> 3;
    const int q = (int)(i & 7);
    if (m >= rows_pad) return;
    u32x4 o4 = {0u, 0u, 0u, 0u};
    if (m < M) {
        const V3 x = load3(xs, m), v = load3(vs, m), n = load3(ns, m);
        const int cx = embed_width(mx), cv = embed_width(mv), c = cx + cv + 3;
        o4 = pack8(q, c, c <= 32, [&](int k) {
            if (k < cx) return embed_col(x, x, k, mx, false);
            if (k < cx + cv) return embed_col(v, v, k - cx, mv, false);
            return pick(n, k - cx - cv);
        });
    }
    out[i] = o4;
}

// rows [0, M): d4 = g_rgb rgb (1 - rgb) (3 columns, split); d4 itself in fp32 (may be NULL) and its sums over the block's 32 rows
// (block_sums [gridDim.x, 3]: the bias gradient is their sum - a [M, 3] -> [3] reduction is ATen's slowest shape, 0.45 ms)
__global__ void __launch_bounds__(256)
k_operand_rgb_delta(const float* __restrict__ rgb, const float* __restrict__ g_rgb, long long M, long long rows_pad, u32x4* __restrict__ out,
                    float* __restrict__ d4, float* __restrict__ block_sums) {
    __shared__ float part[4][3];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long m = i >> 3;
    const int q = (int)(i & 7);
    u32x4 o4 = {0u, 0u, 0u, 0u};
    V3 d = {0.f, 0.f, 0.f};
    if (m < M && (q == 0 || q == 4)) {
        const V3 c = load3(rgb, m), g = load3(g_rgb, m);
        d = V3{g.x * c.x * (1.0f - c.x), g.y * c.y * (1.0f - c.y), g.z * c.z * (1.0f - c.z)};
        if (q == 0 && d4) { d4[3 * m] = d.x; d4[3 * m + 1] = d.y; d4[3 * m + 2] = d.z; }
        o4 = pack8(q, 3, true, [&](int k) { return pick(d, k); });
    }
    if (m < rows_pad) out[i] = o4;
    // the q == 0 lanes of a wave hold 8 rows: sum them (fixed order), then the 4 waves
    float sx = q == 0 ? d.x : 0.f, sy = q == 0 ? d.y : 0.f, sz = q == 0 ? d.z : 0.f;
#pragma unroll
    for (int o = 32; o >= 8; o >>= 1) { sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64); sz += __shfl_xor(sz, o, 64); }
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6][0] = sx; part[threadIdx.x >> 6][1] = sy; part[threadIdx.x >> 6][2] = sz; }
    __syncthreads();
    if (threadIdx.x < 3) block_sums[3 * (size_t)blockIdx.x + threadIdx.x] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// rows [0, M): sbar (1 column, split); rows [rows_pad, 2 rows_pad): 1
__global__ void __launch_bounds__(256)
k_operand_sbar_ones(const float* __restrict__ sbar, long long M, long long rows_pad, u32x4* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long row = i >> 3;
    const int q = (int)(i & 7);
    if (row >= 2 * rows_pad) return;
    u32x4 o4 = {0u, 0u, 0u, 0u};
    if (row >= rows_pad) {
        if (q == 0) o4[0] = 0x3f80u;                          // bf16(1.0) in column 0
    } else if (row < M && (q == 0 || q == 4)) {
        const float s = sbar[row];
        o4 = pack8(q, 1, true, [&](int) { return s; });
    }
    out[i] = o4;
}

static inline dim3 grid8(long long rows) { return dim3((unsigned)((rows * 8 + 255) / 256)); }
static inline int check_rows(long long M, long long rows_pad, const char* who) {
    if (M < 0 || rows_pad < M || rows_pad >= (1ll << 28)) { set_last_error(who); return 2; }
    return 0;
}

}  // namespace p2
}  // namespace nerfart

using namespace nerfart;
using namespace nerfart::p2;

extern "C" {

// pts [R P, 3] = rays_o + rays_dn * depth, view [R P, 3] = rays_dn per sample (view may be NULL)
int nerfart_ray_points(const float* rays_o, const float* rays_dn, const float* depth, long long n_rays, int P, float* pts, float* view,
                       void* stream) {
    if (n_rays <= 0 || P <= 0) return 0;
    if (!rays_o || !rays_dn || !depth || !pts) { set_last_error("ray_points: null argument"); return 2; }
    if (n_rays * P >= (1ll << 31)) { set_last_error("ray_points: n_rays * P must stay below 2^31"); return 2; }
    hipLaunchKernelGGL(k_ray_points, dim3((unsigned)((n_rays * P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_dn, depth,
                       n_rays, P, pts, view);
    NERFART_HIP(hipGetLastError());
    return 0;
}

// VolSDF pass 2, between the radiance net's backward and the second-order SDF sweep.  w_eikonal 0: no eikonal term.
// eik_group_rays: rays per reference patch inside this launch (<= 0: the launch is one patch).  eik_ray [n_rays]: each ray's share of
// the eikonal loss (sum them); g_n / g_n_extra may be NULL (zero); R_bg <= 0: no sphere clamp (NeuS: sbar = g_sdf).
int nerfart_volsdf_pass2_cotangents(const float* pts, const float* sdf, const float* g_sdf, const float* nabla, const float* g_n,
                                    const float* g_n_extra, long long n_rays, int P, float R_bg, float w_eikonal,
                                    long long eik_group_rays, float* sbar, float* nbar, float* eik_ray, void* stream) {
    if (n_rays <= 0 || P <= 0) return 0;
    if (!pts || !sdf || !g_sdf || !nabla || !sbar || !nbar || !eik_ray) { set_last_error("pass2_cotangents: null argument"); return 2; }
    if (n_rays * P >= (1ll << 31)) { set_last_error("pass2_cotangents: n_rays * P must stay below 2^31"); return 2; }
    hipLaunchKernelGGL(k_volsdf_cotangents, dim3((unsigned)n_rays), dim3(64), 0, (hipStream_t)stream, pts, sdf, g_sdf, nabla, g_n,
                       g_n_extra, n_rays, P, R_bg, w_eikonal, eik_group_rays, sbar, nbar, eik_ray);
    NERFART_HIP(hipGetLastError());
    return 0;
}

// out: bf16 [2 rows_pad, 64].  multires < 0: no encoding (3 columns); 3 + 6 multires <= 64.
int nerfart_wgrad_operand_embed_pair(const float* pts, const float* dir, long long M, long long rows_pad, int multires, void* out,
                                     void* stream) {
    if (rows_pad <= 0) return 0;
    if (check_rows(M, rows_pad, "wgrad_operand_embed_pair: need 0 <= M <= rows_pad < 2^28")) return 2;
    if ((M > 0 && (!pts || !dir)) || !out || multires > 10) { set_last_error("wgrad_operand_embed_pair: null argument or multires > 10"); return 2; }
    hipLaunchKernelGGL(k_operand_embed_pair, grid8(2 * rows_pad), dim3(256), 0, (hipStream_t)stream, pts, dir, M, rows_pad, multires,
                       (u32x4*)out);
    NERFART_HIP(hipGetLastError());
    return 0;
}

// out: bf16 [rows_pad, 64] = [embed(x, multires_x) | embed(view, multires_view) | normals]; at most 64 columns
int nerfart_wgrad_operand_inputs(const float* x, int multires_x, const float* view, int multires_view, const float* normals, long long M,
                                 long long rows_pad, void* out, void* stream) {
    if (rows_pad <= 0) return 0;
    if (check_rows(M, rows_pad, "wgrad_operand_inputs: need 0 <= M <= rows_pad < 2^28")) return 2;
    const int c = (multires_x < 0 ? 3 : 3 + 6 * multires_x) + (multires_view < 0 ? 3 : 3 + 6 * multires_view) + 3;
    if ((M > 0 && (!x || !view || !normals)) || !out || c > 64) { set_last_error("wgrad_operand_inputs: null argument or more than 64 columns"); return 2; }
    hipLaunchKernelGGL(k_operand_inputs, grid8(rows_pad), dim3(256), 0, (hipStream_t)stream, x, multires_x, view, multires_view, normals,
                       M, rows_pad, (u32x4*)out);
    NERFART_HIP(hipGetLastError());
    return 0;
}

// out: bf16 [rows_pad, 64] (hi 0..2, lo 32..34); d4: fp32 [M, 3] or NULL; block_sums: fp32 [ceil(rows_pad / 32), 3], the sums of d4 over
// each 32-row block (their sum over the blocks = the bias gradient)
int nerfart_wgrad_operand_rgb_delta(const float* rgb, const float* g_rgb, long long M, long long rows_pad, void* out, float* d4,
                                    float* block_sums, void* stream) {
    if (rows_pad <= 0) return 0;
    if (check_rows(M, rows_pad, "wgrad_operand_rgb_delta: need 0 <= M <= rows_pad < 2^28")) return 2;
    if ((M > 0 && (!rgb || !g_rgb)) || !out || !block_sums) { set_last_error("wgrad_operand_rgb_delta: null argument"); return 2; }
    hipLaunchKernelGGL(k_operand_rgb_delta, grid8(rows_pad), dim3(256), 0, (hipStream_t)stream, rgb, g_rgb, M, rows_pad, (u32x4*)out, d4,
                       block_sums);
    NERFART_HIP(hipGetLastError());
    return 0;
}

// out: bf16 [2 rows_pad, 64]: sbar (hi column 0, lo column 32) in rows [0, M), 1 in column 0 of rows [rows_pad, 2 rows_pad)
int nerfart_wgrad_operand_sbar_ones(const float* sbar, long long M, long long rows_pad, void* out, void* stream) {
    if (rows_pad <= 0) return 0;
    if (check_rows(M, rows_pad, "wgrad_operand_sbar_ones: need 0 <= M <= rows_pad < 2^28")) return 2;
    if ((M > 0 && !sbar) || !out) { set_last_error("wgrad_operand_sbar_ones: null argument"); return 2; }
    hipLaunchKernelGGL(k_operand_sbar_ones, grid8(2 * rows_pad), dim3(256), 0, (hipStream_t)stream, sbar, M, rows_pad, (u32x4*)out);
    NERFART_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
