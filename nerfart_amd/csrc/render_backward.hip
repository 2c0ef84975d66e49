// render_backward.hip - the RAY-LEVEL backward of the renderer (boundary B1 "bwd"; row a19).
//
// What the reference does with `rgb.backward(gradient)` + `eikonal.backward()` through render_fn for one patch of rays
// (models/frameworks/volsdf.py:759-770, neus.py:520-576) is ONE call here:
//
//   nerfart_volsdf_render_bwd / nerfart_neus_render_bwd
//       rays, the sample depths pass 1 drew, d loss / d rgb (+ optional d loss / d acc, extra nabla cotangents), optionally the per-sample
//       state pass 1 kept (sdf, grad sdf, layer-7 activation: same weights => identical values) ->
//       the RAW parameter-gradient buffer (nerfart_pass2_raw_layout), accumulated into: the fp32 results of the weight-gradient
//       reductions in the kernels' unit order, the bias column sums, d loss / d alpha, d beta (VolSDF) or d s (NeuS), the eikonal loss.
//   nerfart_sdf_param_bwd
//       the SDF net's share on its own: parameter gradients of  sbar . sdf + hbar7 . h7 + nbar . grad_x sdf  at given points
//       (the reconstruction objective's free eikonal points, volsdf.py:799-806)
//   nerfart_fold_weight_grads
//       raw buffer -> gradients of the FOLDED weight matrices / biases in the reference's feature order (un-permutation of the unit
//       order, hi + lo halves of the narrow operands, the dumps' 1/65535 scale, the skip layer's 1/sqrt(2)); linear, so it runs once
//       per step on the sum over all launch groups
//   nerfart_weight_norm_bwd
//       nn.utils.weight_norm's chain rule: d loss / d W -> d loss / d weight_g, d loss / d weight_v (models/base.py:226-227)
//
// The sequence inside (every stage is an entry point of its own in include/nerfart_hip.h, exercised one by one by the tests):
//   normalise dirs -> sample points -> [SDF + grad SDF + h7 unless kept] -> radiance forward with activation dumps -> compositor
//   backward -> radiance backward (deltas, g_h7, g_n) -> cotangents of the second-order SDF sweep (sphere clamp mask, eikonal
//   gradient with the per-patch mean) -> k_sdf_fwd2 / k_sdf_bwd2 -> the weight-gradient reductions over the point-major dumps.
// Everything runs on the caller's stream out of ONE caller-owned workspace; nothing is allocated, nothing synchronises.
#include "nerfart_common.h"
#include "../../include/nerfart_hip.h"
#include <string>

namespace nerfart {
namespace rbwd {

// ---- raw buffer sections (floats); every size is a multiple of 4 floats ----------------------------------------------------------
enum { S_SURF_WW, S_SURF_WE, S_SURF_CS0, S_SURF_CS17, S_SURF_W8, S_SURF_B8, S_RAD_WW, S_RAD_CS, S_RAD_W4, S_RAD_B4, S_RAD_WEX, S_RAD_WH7,
       S_SCALARS, N_SECTIONS };
static const long long kSectionFloats[N_SECTIONS] = {
    7 * 65536,      // SURF_WW  [7][256][256]  65535 * (zbar_l^T a_{l-1} + (t_l d_l)^T adot_{l-1}), l = 1..7
    2 * 16384,      // SURF_WE  [2][256][64]   the same against the encoding pair, layers 0 and 4
    2 * 256,        // SURF_CS0 [2][256]       65535 * column sums of zbar_0 (row 0; row 1 = zbar_4 again, unused)
    7 * 256,        // SURF_CS17[7][256]       65535 * column sums of zbar_1..7
    16384,          // SURF_W8  [256][64]      a7^T sbar + column sums of adot7 (hi column 0, lo column 32)
    4,              // SURF_B8  [1] (+3 pad)   sum of sbar
    4 * 65536,      // RAD_WW   [4][256][256]  delta_l^T act_{l-1}: (d0, f), (d1, r0), (d2, r1), (d3, r2)
    5 * 256,        // RAD_CS   [5][256]       column sums of d0..d3 and of the geometry-feature cotangent
    16384,          // RAD_W4   [256][64]      r3^T d4 (hi columns 0..2, lo 32..34)
    4,              // RAD_B4   [3] (+1 pad)   column sums of d4
    16384,          // RAD_WEX  [256][64]      d0^T [x | v | n]
    65536,          // RAD_WH7  [256][256]     (geometry-feature cotangent)^T a7: rows 1.. of the last SDF layer
    4,              // SCALARS  d loss / d alpha, d loss / d beta, d loss / d s, sum of the patches' eikonal losses
};
static long long section_offset(int s) {
    long long o = 0;
    for (int i = 0; i < s; ++i) o += kSectionFloats[i];
    return o;
}

static inline int embed_width(int multires) { return multires < 0 ? 3 : 3 + 6 * multires; }

// ---- small kernels ---------------------------------------------------------------------------------------------------------------
// two-stage deterministic column sums of x [n][C] (C <= 4): partial[b][c], then out[c] += sum_b partial[b][c]
template <int C>
__global__ void __launch_bounds__(256) k_partial_sums(const float* __restrict__ x, long long n, float* __restrict__ partial) {
    __shared__ float red[4][C];
    float s[C];
#pragma unroll
    for (int c = 0; c < C; ++c) s[c] = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
#pragma unroll
        for (int c = 0; c < C; ++c) s[c] += x[i * C + c];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[c] += __shfl_xor(s[c], o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) red[threadIdx.x >> 6][c] = s[c];
    }
    __syncthreads();
    if (threadIdx.x < C) partial[(size_t)blockIdx.x * C + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
template <int C>
__global__ void __launch_bounds__(64) k_final_sums(const float* __restrict__ partial, int nb, float* __restrict__ out) {
    float s[C];
#pragma unroll
    for (int c = 0; c < C; ++c) s[c] = 0.f;
    for (int b = threadIdx.x; b < nb; b += 64) {
#pragma unroll
        for (int c = 0; c < C; ++c) s[c] += partial[(size_t)b * C + c];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[c] += __shfl_xor(s[c], o, 64);
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) out[c] += s[c];
    }
}
static constexpr int kSumBlocks = 256;
template <int C>
static int add_sums(const float* x, long long n, float* partial, float* out, hipStream_t st) {
    if (n <= 0) return 0;
    const int nb = (int)((n + 1023) / 1024 < kSumBlocks ? (n + 1023) / 1024 : kSumBlocks);
    hipLaunchKernelGGL(k_partial_sums<C>, dim3(nb), dim3(256), 0, st, x, n, partial);
    hipLaunchKernelGGL(k_final_sums<C>, dim3(1), dim3(64), 0, st, (const float*)partial, nb, out);
    NERFART_HIP(hipGetLastError());
    return 0;
}

// NeuS: the radiance samples sit at the interval mid-points (neus.py:343-345)
__global__ void __launch_bounds__(256) k_mid_depths(const float* __restrict__ d, long long R, int P, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * (P - 1)) return;
    const long long r = i / (P - 1);
    const int p = (int)(i - r * (P - 1));
    out[i] = 0.5f * (d[r * P + p + 1] + d[r * P + p]);
}

// position of feature f in the kernels' unit order (nerfart_amd/packing.py: unit_feature_hidden, inverted)
__device__ __forceinline__ int unit_pos(int f) {
    const int u = f >> 5, r = f & 31;
    return r < 16 ? 32 * u + 8 * (r >> 2) + (r & 3) : 32 * u + 8 * ((r - 16) >> 2) + 4 + ((r - 16) & 3);
}

// the layer-7 activation h7 [M, 256] fp32 (feature order) -> [rows_pad, 256] bf16 in unit order, rows M.. zero: the operand
// k_sdf_fwd2_bf16 dumps (slot 7, value rows) when the SDF sweep runs; built here when only the radiance half is asked for
__global__ void __launch_bounds__(256) k_h7_units(const float* __restrict__ h7, long long M, long long rows_pad, unsigned short* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long m = i >> 8;
    const int p = (int)(i & 255);
    if (m >= rows_pad) return;
    unsigned short b = 0;
    if (m < M) {
        const int u = p >> 5, g = (p >> 3) & 3, e = p & 7;
        const int f = 32 * u + (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4));
        const unsigned x = __float_as_uint(h7[m * 256 + f]);
        b = (unsigned short)((x + 0x7fffu + ((x >> 16) & 1u)) >> 16);       // round to nearest even, as torch's .to(torch.bfloat16)
    }
    out[i] = b;
}

struct FoldLayer { long long w_off, b_off; int out, in, kind, idx; };
// kind: 0 surf layer 0 (encoding), 1 surf hidden, 2 surf skip layer, 3 surf last layer (sdf row + geometry-feature rows),
//       4 rad layer 0, 5 rad hidden, 6 rad last
struct FoldArgs { FoldLayer L[14]; int n_layers; int nenc, nex; long long sec[N_SECTIONS]; };

__global__ void __launch_bounds__(256) k_fold(const float* __restrict__ raw, FoldArgs a, float* __restrict__ out) {
    const int li = blockIdx.y;
    if (li >= a.n_layers) return;
    const FoldLayer L = a.L[li];
    const long long n_w = (long long)L.out * L.in;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_w + L.out) return;
    const float sc = 1.0f / 65535.0f, rs2 = 0.70710678118654752440f;
    const float* ww = raw + a.sec[S_SURF_WW];
    const float* we = raw + a.sec[S_SURF_WE];
    const float* rw = raw + a.sec[S_RAD_WW];
    if (e >= n_w) {                                     // bias entry
        const int o = (int)(e - n_w);
        float v;
        switch (L.kind) {
            case 0: v = raw[a.sec[S_SURF_CS0] + unit_pos(o)] * sc; break;
            case 1: case 2: v = raw[a.sec[S_SURF_CS17] + (L.idx - 1) * 256 + unit_pos(o)] * sc; break;
            case 3: v = o == 0 ? raw[a.sec[S_SURF_B8]] : raw[a.sec[S_RAD_CS] + 4 * 256 + unit_pos(o - 1)]; break;
            case 4: case 5: v = raw[a.sec[S_RAD_CS] + L.idx * 256 + unit_pos(o)]; break;
            default: v = raw[a.sec[S_RAD_B4] + o]; break;
        }
        out[L.b_off + o] = v;
        return;
    }
    const int o = (int)(e / L.in), i = (int)(e - (long long)o * L.in);
    float v;
    switch (L.kind) {
        case 0: {
            const float* p = we + (size_t)unit_pos(o) * 64;
            v = (p[i] + (a.nenc <= 32 ? p[32 + i] : 0.f)) * sc;
        } break;
        case 1: v = ww[(size_t)(L.idx - 1) * 65536 + (size_t)unit_pos(o) * 256 + unit_pos(i)] * sc; break;
        case 2: {
            const int hw = 256 - a.nenc;
            if (i < hw) v = ww[(size_t)(L.idx - 1) * 65536 + (size_t)unit_pos(o) * 256 + unit_pos(i)] * (sc * rs2);
            else {
                const float* p = we + 16384 + (size_t)unit_pos(o) * 64;
                const int c = i - hw;
                v = (p[c] + (a.nenc <= 32 ? p[32 + c] : 0.f)) * (sc * rs2);
            }
        } break;
        case 3: {
            if (o == 0) { const float* p = raw + a.sec[S_SURF_W8] + (size_t)unit_pos(i) * 64; v = p[0] + p[32]; }
            else v = raw[a.sec[S_RAD_WH7] + (size_t)unit_pos(o - 1) * 256 + unit_pos(i)];
        } break;
        case 4: {
            if (i < a.nex) { const float* p = raw + a.sec[S_RAD_WEX] + (size_t)unit_pos(o) * 64; v = p[i] + (a.nex <= 32 ? p[32 + i] : 0.f); }
            else v = rw[(size_t)unit_pos(o) * 256 + unit_pos(i - a.nex)];
        } break;
        case 5: v = rw[(size_t)L.idx * 65536 + (size_t)unit_pos(o) * 256 + unit_pos(i)]; break;
        default: { const float* p = raw + a.sec[S_RAD_W4] + (size_t)unit_pos(i) * 64; v = p[o] + p[32 + o]; } break;
    }
    out[L.w_off + e] = v;
}

// one wave per output row: g_g[o] (+)= (dW_o . v_o) / |v_o|,  g_v[o] (+)= g_o / |v_o| (dW_o - v_o (dW_o . v_o) / |v_o|^2)
__global__ void __launch_bounds__(64) k_weight_norm_bwd(const float* __restrict__ dW, const float* __restrict__ v, const float* __restrict__ g, int in,
                                                        float* __restrict__ g_v, float* __restrict__ g_g, int accumulate) {
    const int o = blockIdx.x, lane = threadIdx.x;
    const float* dw = dW + (size_t)o * in;
    const float* vv = v + (size_t)o * in;
    float dot = 0.f, n2 = 0.f;
    for (int i = lane; i < in; i += 64) { dot = fmaf(dw[i], vv[i], dot); n2 = fmaf(vv[i], vv[i], n2); }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) { dot += __shfl_xor(dot, s, 64); n2 += __shfl_xor(n2, s, 64); }
    const float nrm = sqrtf(n2), inv = 1.0f / nrm;
    const float gg = dot * inv;
    if (lane == 0 && g_g) g_g[o] = accumulate ? g_g[o] + gg : gg;
    if (g_v) {
        const float k1 = g[o] * inv, k2 = dot / n2;
        for (int i = lane; i < in; i += 64) {
            const float t = k1 * (dw[i] - vv[i] * k2);
            g_v[(size_t)o * in + i] = accumulate ? g_v[(size_t)o * in + i] + t : t;
        }
    }
}

// ---- workspace carving -----------------------------------------------------------------------------------------------------------
struct Carver {
    char* base; size_t off;
    explicit Carver(void* p) : base((char*)p), off(0) {}
    template <class T> T* take(size_t n) { T* r = base ? (T*)(base + off) : nullptr; off += (n * sizeof(T) + 255) & ~(size_t)255; return r; }
    void* bytes(size_t n) { return take<char>(n); }
};
static inline long long up(long long m, long long k) { return (m + k - 1) / k * k; }
static inline long long max3(long long a, long long b, long long c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

// scratch of the SDF net's share: dumps, the two narrow operands, zero cotangents when an input is NULL, partial sums, split-K partials
struct SurfWs { void *f2, *r2, *E2, *SO; float *zeros_h7, *zeros_s, *partial; void* wg; long long wg_bytes; };
// zero_h7 / zero_s: the caller will pass a NULL hbar7 / sbar cotangent (VolSDF always supplies both: no 1 KiB per point of zeros carved)
static SurfWs carve_surf(Carver& c, long long M, bool zero_h7 = true, bool zero_s = true) {
    SurfWs w;
    const long long Mp = up(M, 64);
    w.f2 = c.bytes((size_t)nerfart_sdf_fwd2_dump_bytes(M));
    w.r2 = c.bytes((size_t)nerfart_sdf_bwd2_dump_bytes(M));
    w.E2 = c.bytes((size_t)2 * Mp * 128);
    w.SO = c.bytes((size_t)2 * Mp * 128);
    w.zeros_h7 = zero_h7 ? c.take<float>((size_t)M * 256) : nullptr;
    w.zeros_s = zero_s ? c.take<float>((size_t)M) : nullptr;
    w.partial = c.take<float>((size_t)kSumBlocks * 4);
    w.wg_bytes = max3(nerfart_wgrad_workspace_bytes(7, 2 * Mp, 256), nerfart_wgrad_workspace_bytes(2, 2 * Mp, 64), nerfart_wgrad_workspace_bytes(1, 2 * Mp, 64));
    w.wg = c.bytes((size_t)w.wg_bytes);
    return w;
}

// k_sdf_fwd2 / k_sdf_bwd2 + the reductions over their dumps, accumulated into raw.  *a7 = the layer-7 activation (value rows of the
// forward dump's slot 7: [Mp, 256] bf16 in unit order) for the geometry-feature rows of the last layer.
static int surf_param_bwd(const float* surf_blob, int multires, const float* pts, long long M, const float* sbar, const float* hbar7, const float* nbar,
                          float* raw, const SurfWs& w, const void** a7, hipStream_t st) {
    const long long Mp = up(M, 64);
    const long long slot = 2 * Mp * 512;                                   // bytes between the dumps' slots
    if ((!hbar7 && !w.zeros_h7) || (!sbar && !w.zeros_s)) { set_last_error("surf_param_bwd: NULL cotangent without a carved zero buffer (internal)"); return 3; }
    if (!hbar7) { NERFART_HIP(hipMemsetAsync(w.zeros_h7, 0, (size_t)M * 256 * 4, st)); hbar7 = w.zeros_h7; }
    if (!sbar) { NERFART_HIP(hipMemsetAsync(w.zeros_s, 0, (size_t)M * 4, st)); sbar = w.zeros_s; }
    if (int rc = nerfart_sdf_fwd2(surf_blob, pts, nbar, M, w.f2, st)) return rc;
    if (int rc = nerfart_sdf_bwd2(surf_blob, M, hbar7, sbar, w.f2, w.r2, st)) return rc;
    const char* RZ = (const char*)w.r2;                                    // [8][2 Mp][256] bf16: 65535 * [zbar_l; t_l d_l]
    const char* FA = (const char*)w.f2;                                    // [8][2 Mp][256] bf16: [a_l; adot_l]  (+ 8 softplus' slots)
    // layers 1..7 against the previous layer's (a | adot); column sums of zbar_1..7 (the first Mp rows) in the same pass
    if (int rc = nerfart_wgrad_bf16(RZ + slot, slot, FA, slot, 7, 2 * Mp, 256, Mp, raw + section_offset(S_SURF_WW), raw + section_offset(S_SURF_CS17), 1,
                                    w.wg, w.wg_bytes, st)) return rc;
    // layers 0 and 4 (four slots apart) against the encoding pair [e; edot]; column sums of zbar_0
    if (int rc = nerfart_wgrad_operand_embed_pair(pts, nbar, M, Mp, multires, w.E2, st)) return rc;
    if (int rc = nerfart_wgrad_bf16(RZ, 4 * slot, w.E2, 0, 2, 2 * Mp, 64, Mp, raw + section_offset(S_SURF_WE), raw + section_offset(S_SURF_CS0), 1,
                                    w.wg, w.wg_bytes, st)) return rc;
    // the sdf row of the last layer: a7^T sbar + column sums of adot7 = FA[7]^T [sbar; 1]
    if (int rc = nerfart_wgrad_operand_sbar_ones(sbar, M, Mp, w.SO, st)) return rc;
    if (int rc = nerfart_wgrad_bf16(FA + 7 * slot, 0, w.SO, 0, 1, 2 * Mp, 64, 0, raw + section_offset(S_SURF_W8), nullptr, 1, w.wg, w.wg_bytes, st)) return rc;
    if (int rc = add_sums<1>(sbar, M, w.partial, raw + section_offset(S_SURF_B8), st)) return rc;
    if (a7) *a7 = FA + 7 * slot;
    return 0;
}

// scratch of the radiance net's share
struct RadWs { float *rgb, *g_rad, *g_h7, *g_n, *block_sums, *partial; void *dump, *bdump, *D4, *EX; void* wg; long long wg_bytes; };
static RadWs carve_rad(Carver& c, long long M) {
    RadWs w;
    const long long Mp = up(M, 128);
    w.rgb = c.take<float>((size_t)M * 3);
    w.g_rad = c.take<float>((size_t)M * 3);
    w.g_h7 = c.take<float>((size_t)M * 256);
    w.g_n = c.take<float>((size_t)M * 3);
    w.dump = c.bytes((size_t)nerfart_radiance_dump_bytes(M));
    w.bdump = c.bytes((size_t)nerfart_radiance_dump_bytes(M));
    w.D4 = c.bytes((size_t)Mp * 128);
    w.EX = c.bytes((size_t)Mp * 128);
    w.block_sums = c.take<float>((size_t)((Mp + 31) / 32) * 3);
    w.partial = c.take<float>((size_t)kSumBlocks * 4);
    w.wg_bytes = max3(nerfart_wgrad_workspace_bytes(4, Mp, 256), nerfart_wgrad_workspace_bytes(1, Mp, 256), nerfart_wgrad_workspace_bytes(1, Mp, 64));
    w.wg = c.bytes((size_t)w.wg_bytes);
    return w;
}

// reductions of the radiance deltas against the activations (after nerfart_radiance_fwd_dump / _bwd filled the dumps).  a7: the
// layer-7 activation in unit order, >= up(M, 64) rows.  train_radiance == 0: only the rows of the last SDF layer (frozen radiance net,
// neus.py:455-456).
static int rad_param_bwd(int multires_view, const float* x, const float* v, const float* n, long long M, const float* g_rad, const void* a7, int train_radiance,
                         float* raw, const RadWs& w, hipStream_t st) {
    const long long Mp = up(M, 128), Mp64 = up(M, 64);
    const long long slot = Mp * 512;
    const char* acts = (const char*)w.dump;                                // f, r0, r1, r2, r3    [5][Mp][256] bf16
    const char* deltas = (const char*)w.bdump;                             // d0, d1, d2, d3, g_f  (0 for the padded points)
    if (int rc = nerfart_wgrad_bf16(deltas + 4 * slot, 0, a7, 0, 1, Mp64, 256, Mp64, raw + section_offset(S_RAD_WH7), raw + section_offset(S_RAD_CS) + 4 * 256, 1,
                                    w.wg, w.wg_bytes, st)) return rc;
    if (!train_radiance) return 0;
    if (int rc = nerfart_wgrad_bf16(deltas, slot, acts, slot, 4, Mp, 256, Mp, raw + section_offset(S_RAD_WW), raw + section_offset(S_RAD_CS), 1, w.wg, w.wg_bytes, st)) return rc;
    if (int rc = nerfart_wgrad_operand_rgb_delta(w.rgb, g_rad, M, Mp, w.D4, nullptr, w.block_sums, st)) return rc;
    if (int rc = nerfart_wgrad_bf16(acts + 4 * slot, 0, w.D4, 0, 1, Mp, 64, 0, raw + section_offset(S_RAD_W4), nullptr, 1, w.wg, w.wg_bytes, st)) return rc;
    if (int rc = add_sums<3>(w.block_sums, (Mp + 31) / 32, w.partial, raw + section_offset(S_RAD_B4), st)) return rc;
    if (int rc = nerfart_wgrad_operand_inputs(x, -1, v, multires_view, n, M, Mp, w.EX, st)) return rc;
    if (int rc = nerfart_wgrad_bf16(deltas, 0, w.EX, 0, 1, Mp, 64, 0, raw + section_offset(S_RAD_WEX), nullptr, 1, w.wg, w.wg_bytes, st)) return rc;
    return 0;
}

static int view_multires(int view_tiles, int* mv) {
    if (view_tiles == 1) { *mv = -1; return 0; }
    if (view_tiles == 3) { *mv = 4; return 0; }
    set_last_error("render_bwd: view_tiles must be 1 (raw view dirs) or 3 (multires_view = 4)");
    return 2;
}

static constexpr long long kMaxPoints = 1ll << 21;

// ---- VolSDF ---------------------------------------------------------------------------------------------------------------------
struct VolWs { float *dn, *pts, *view, *sdf, *nab, *h7, *g_sdf, *sbar, *nbar, *eik_ray, *partial; void* nabla_ws; long long nabla_ws_bytes; RadWs rad; SurfWs surf; };
static VolWs carve_volsdf(Carver& c, long long R, int P, int have_state) {
    VolWs w;
    const long long M = R * P;
    w.dn = c.take<float>((size_t)R * 3);
    w.pts = c.take<float>((size_t)M * 3);
    w.view = c.take<float>((size_t)M * 3);
    w.sdf = w.nab = w.h7 = nullptr; w.nabla_ws = nullptr; w.nabla_ws_bytes = 0;
    if (!have_state) {
        w.sdf = c.take<float>((size_t)M);
        w.nab = c.take<float>((size_t)M * 3);
        w.h7 = c.take<float>((size_t)M * 256);
        w.nabla_ws_bytes = nerfart_sdf_nabla_workspace_bytes(1);
        w.nabla_ws = c.bytes((size_t)w.nabla_ws_bytes);
    }
    w.g_sdf = c.take<float>((size_t)M);
    w.sbar = c.take<float>((size_t)M);
    w.nbar = c.take<float>((size_t)M * 3);
    w.eik_ray = c.take<float>((size_t)R);
    w.partial = c.take<float>((size_t)kSumBlocks * 4);
    w.rad = carve_rad(c, M);
    w.surf = carve_surf(c, M, false, false);                               // sbar and hbar7 (= g_h7 of the radiance net) are always supplied
    return w;
}

// ---- NeuS -----------------------------------------------------------------------------------------------------------------------
struct NeusWs { float *dn, *pts, *d_mid, *pts_m, *view_m, *sdf, *nab, *sdf_m, *nab_m, *h7_m, *g_sdf, *sbar, *nbar, *eik_ray, *partial; void* nabla_ws;
                long long nabla_ws_bytes; RadWs rad; SurfWs surf; };
static NeusWs carve_neus(Carver& c, long long R, int P, int have_state) {
    NeusWs w;
    const long long M = R * P, Mm = R * (P - 1);
    w.dn = c.take<float>((size_t)R * 3);
    w.pts = c.take<float>((size_t)M * 3);
    w.d_mid = c.take<float>((size_t)Mm);
    w.pts_m = c.take<float>((size_t)Mm * 3);
    w.view_m = c.take<float>((size_t)Mm * 3);
    w.sdf = w.nab = nullptr;
    if (!have_state) { w.sdf = c.take<float>((size_t)M); w.nab = c.take<float>((size_t)M * 3); }
    w.sdf_m = c.take<float>((size_t)Mm);
    w.nab_m = c.take<float>((size_t)Mm * 3);
    w.h7_m = c.take<float>((size_t)Mm * 256);
    w.nabla_ws_bytes = nerfart_sdf_nabla_workspace_bytes(1);
    w.nabla_ws = c.bytes((size_t)w.nabla_ws_bytes);
    w.g_sdf = c.take<float>((size_t)M);
    w.sbar = c.take<float>((size_t)M);
    w.nbar = c.take<float>((size_t)M * 3);
    w.eik_ray = c.take<float>((size_t)R);
    w.partial = c.take<float>((size_t)kSumBlocks * 4);
    w.rad = carve_rad(c, Mm);
    w.surf = carve_surf(c, M);                                             // the samples' sweep, then the mid-points' (M >= Mm), one after the other
    return w;
}

static int check_rays(long long R, int P, const char* who) {
    if (P < 2 || P > 513) { set_last_error((std::string(who) + ": 2 <= P <= 513").c_str()); return 2; }
    if (R * P > kMaxPoints) { set_last_error((std::string(who) + ": at most 2^21 sample points per call (n_rays * P)").c_str()); return 2; }
    return 0;
}

}  // namespace rbwd
}  // namespace nerfart

using namespace nerfart;
using namespace nerfart::rbwd;

extern "C" {

long long nerfart_pass2_raw_layout(long long* offsets) {
    long long o = 0;
    for (int s = 0; s < N_SECTIONS; ++s) {
        if (offsets) offsets[s] = o;
        o += kSectionFloats[s];
    }
    if (offsets) offsets[N_SECTIONS] = o;
    return o;
}

long long nerfart_sdf_param_bwd_workspace_bytes(long long M) {
    if (M <= 0) return 0;
    Carver c(nullptr);
    carve_surf(c, M);
    return (long long)c.off;
}

int nerfart_sdf_param_bwd(const float* surf_blob, int multires, const float* pts, long long M, const float* sbar, const float* hbar7, const float* nbar,
                          float* raw, void* workspace, long long workspace_bytes, void* stream) {
    if (int rc_ = blob_term_check(surf_blob, 1, "nerfart_sdf_param_bwd")) return rc_;
    if (M <= 0) return 0;
    if (M > kMaxPoints) { set_last_error("sdf_param_bwd: at most 2^21 points per call"); return 2; }
    if (!surf_blob || !pts || !nbar || !raw) { set_last_error("sdf_param_bwd: null argument (sbar / hbar7 may be NULL = zero; nbar may not)"); return 2; }
    if (!workspace || workspace_bytes < nerfart_sdf_param_bwd_workspace_bytes(M)) { set_last_error("sdf_param_bwd: workspace missing or smaller than nerfart_sdf_param_bwd_workspace_bytes()"); return 2; }
    Carver c(workspace);
    SurfWs w = carve_surf(c, M);
    return surf_param_bwd(surf_blob, multires, pts, M, sbar, hbar7, nbar, raw, w, nullptr, (hipStream_t)stream);
}

// the radiance net's share on its own (+ the geometry-feature rows of the last SDF layer): forward with dumps, backward, reductions
long long nerfart_radiance_param_bwd_workspace_bytes(long long M) {
    if (M <= 0) return 0;
    Carver c(nullptr);
    carve_rad(c, M);
    c.bytes((size_t)up(M, 64) * 512);
    return (long long)c.off;
}

int nerfart_radiance_param_bwd(const float* rad_blob, int view_tiles, const float* pts, const float* view, const float* nabla, const float* h7,
                               long long M, const float* g_rgb, float* rgb_out, float* g_h7_out, float* g_n_out, int train_radiance, float* raw,
                               void* workspace, long long workspace_bytes, void* stream) {
    if (int rc_ = blob_term_check(rad_blob, 1, "nerfart_radiance_param_bwd")) return rc_;
    if (M <= 0) return 0;
    if (M > kMaxPoints) { set_last_error("radiance_param_bwd: at most 2^21 points per call"); return 2; }
    int mv;
    if (int rc = view_multires(view_tiles, &mv)) return rc;
    if (!rad_blob || !pts || !view || !nabla || !h7 || !g_rgb || !raw) { set_last_error("radiance_param_bwd: null argument"); return 2; }
    if (!workspace || workspace_bytes < nerfart_radiance_param_bwd_workspace_bytes(M)) { set_last_error("radiance_param_bwd: workspace missing or smaller than nerfart_radiance_param_bwd_workspace_bytes()"); return 2; }
    hipStream_t st = (hipStream_t)stream;
    Carver c(workspace);
    RadWs w = carve_rad(c, M);
    const long long Mp64 = up(M, 64);
    void* a7 = c.bytes((size_t)Mp64 * 512);
    if (int rc = nerfart_radiance_fwd_dump(rad_blob, view_tiles, pts, view, M, nabla, h7, w.rgb, w.dump, st)) return rc;
    if (int rc = nerfart_radiance_bwd(rad_blob, M, w.rgb, g_rgb, w.dump, w.bdump, w.g_h7, w.g_n, st)) return rc;
    hipLaunchKernelGGL(k_h7_units, dim3((unsigned)(Mp64)), dim3(256), 0, st, h7, M, Mp64, (unsigned short*)a7);
    NERFART_HIP(hipGetLastError());
    if (int rc = rad_param_bwd(mv, pts, view, nabla, M, g_rgb, a7, train_radiance, raw, w, st)) return rc;
    if (rgb_out) NERFART_HIP(hipMemcpyAsync(rgb_out, w.rgb, (size_t)M * 12, hipMemcpyDeviceToDevice, st));
    if (g_h7_out) NERFART_HIP(hipMemcpyAsync(g_h7_out, w.g_h7, (size_t)M * 1024, hipMemcpyDeviceToDevice, st));
    if (g_n_out) NERFART_HIP(hipMemcpyAsync(g_n_out, w.g_n, (size_t)M * 12, hipMemcpyDeviceToDevice, st));
    return 0;
}

long long nerfart_volsdf_render_bwd_workspace_bytes(int n_rays, int P, int have_state) {
    if (n_rays <= 0 || P < 2) return 0;
    Carver c(nullptr);
    carve_volsdf(c, n_rays, P, have_state);
    return (long long)c.off;
}

int nerfart_volsdf_render_bwd(const float* surf_blob, const float* rad_blob, int view_tiles, int multires, const float* rays_o, const float* rays_d,
                              int n_rays, int P, const float* d_all, const float* g_rgb, const float* g_acc, const float* g_n_extra,
                              const float* sdf_state, const float* nabla_state, const float* h7_state, float R_bg, float alpha, float beta,
                              int white_bkgd, float w_eikonal, int eik_group_rays, int train_radiance, float* raw, void* workspace,
                              long long workspace_bytes, void* stream) {
    if (int rc_ = blob_term_check(surf_blob, 1, "nerfart_volsdf_render_bwd")) return rc_;
    if (int rc_ = blob_term_check(rad_blob, 1, "nerfart_volsdf_render_bwd")) return rc_;
    if (n_rays <= 0) return 0;
    if (int rc = check_rays(n_rays, P, "volsdf_render_bwd")) return rc;
    int mv;
    if (int rc = view_multires(view_tiles, &mv)) return rc;
    if (!surf_blob || !rad_blob || !rays_o || !rays_d || !d_all || !g_rgb || !raw) { set_last_error("volsdf_render_bwd: null argument"); return 2; }
    const int have_state = sdf_state || nabla_state || h7_state;
    if (have_state && !(sdf_state && nabla_state && h7_state)) { set_last_error("volsdf_render_bwd: pass all of sdf_state / nabla_state / h7_state or none"); return 2; }
    if (!workspace || workspace_bytes < nerfart_volsdf_render_bwd_workspace_bytes(n_rays, P, have_state)) {
        set_last_error("volsdf_render_bwd: workspace missing or smaller than nerfart_volsdf_render_bwd_workspace_bytes()"); return 2;
    }
    hipStream_t st = (hipStream_t)stream;
    Carver c(workspace);
    VolWs w = carve_volsdf(c, n_rays, P, have_state);
    const long long R = n_rays, M = R * P;
    float* scal = raw + section_offset(S_SCALARS);
    if (int rc = nerfart_normalize_dirs(rays_d, w.dn, n_rays, st)) return rc;
    if (int rc = nerfart_ray_points(rays_o, w.dn, d_all, R, P, w.pts, w.view, st)) return rc;
    const float *sdf = sdf_state, *nab = nabla_state, *h7 = h7_state;
    if (!have_state) {
        if (int rc = nerfart_sdf_nabla_fwd(surf_blob, 1, w.pts, M, R_bg, w.sdf, w.nab, w.h7, w.nabla_ws, w.nabla_ws_bytes, st)) return rc;
        sdf = w.sdf; nab = w.nab; h7 = w.h7;
    }
    if (int rc = nerfart_radiance_fwd_dump(rad_blob, view_tiles, w.pts, w.view, M, nab, h7, w.rad.rgb, w.rad.dump, st)) return rc;
    if (int rc = nerfart_volsdf_composite_bwd(n_rays, P, d_all, sdf, w.rad.rgb, alpha, beta, white_bkgd, g_rgb, g_acc, w.g_sdf, w.rad.g_rad, scal, st)) return rc;
    if (int rc = nerfart_radiance_bwd(rad_blob, M, w.rad.rgb, w.rad.g_rad, w.rad.dump, w.rad.bdump, w.rad.g_h7, w.rad.g_n, st)) return rc;
    // sbar: no gradient to the net where sdf = min(net, R - |x|) took the sphere; nbar = g_n + the eikonal term's gradient
    if (int rc = nerfart_volsdf_pass2_cotangents(w.pts, sdf, w.g_sdf, nab, w.rad.g_n, g_n_extra, R, P, R_bg, w_eikonal, eik_group_rays, w.sbar, w.nbar, w.eik_ray, st)) return rc;
    if (int rc = add_sums<1>(w.eik_ray, R, w.partial, scal + 3, st)) return rc;
    const void* a7 = nullptr;
    if (int rc = surf_param_bwd(surf_blob, multires, w.pts, M, w.sbar, w.rad.g_h7, w.nbar, raw, w.surf, &a7, st)) return rc;
    return rad_param_bwd(mv, w.pts, w.view, nab, M, w.rad.g_rad, a7, train_radiance, raw, w.rad, st);
}

long long nerfart_neus_render_bwd_workspace_bytes(int n_rays, int P, int have_state) {
    if (n_rays <= 0 || P < 2) return 0;
    Carver c(nullptr);
    carve_neus(c, n_rays, P, have_state);
    return (long long)c.off;
}

int nerfart_neus_render_bwd(const float* surf_blob, const float* rad_blob, int view_tiles, int multires, const float* rays_o, const float* rays_d,
                            int n_rays, int P, const float* d_all, const float* g_rgb, const float* g_acc, const float* sdf_state,
                            const float* nabla_state, float s, int white_bkgd, float w_eikonal, int eik_group_rays, int train_radiance, float* raw,
                            void* workspace, long long workspace_bytes, void* stream) {
    if (int rc_ = blob_term_check(surf_blob, 1, "nerfart_neus_render_bwd")) return rc_;
    if (int rc_ = blob_term_check(rad_blob, 1, "nerfart_neus_render_bwd")) return rc_;
    if (n_rays <= 0) return 0;
    if (int rc = check_rays(n_rays, P, "neus_render_bwd")) return rc;
    int mv;
    if (int rc = view_multires(view_tiles, &mv)) return rc;
    if (!surf_blob || !rad_blob || !rays_o || !rays_d || !d_all || !g_rgb || !raw) { set_last_error("neus_render_bwd: null argument"); return 2; }
    const int have_state = sdf_state || nabla_state;
    if (have_state && !(sdf_state && nabla_state)) { set_last_error("neus_render_bwd: pass both of sdf_state / nabla_state or neither"); return 2; }
    if (!workspace || workspace_bytes < nerfart_neus_render_bwd_workspace_bytes(n_rays, P, have_state)) {
        set_last_error("neus_render_bwd: workspace missing or smaller than nerfart_neus_render_bwd_workspace_bytes()"); return 2;
    }
    hipStream_t st = (hipStream_t)stream;
    Carver c(workspace);
    NeusWs w = carve_neus(c, n_rays, P, have_state);
    const long long R = n_rays, M = R * P, Mm = R * (P - 1);
    float* scal = raw + section_offset(S_SCALARS);
    if (int rc = nerfart_normalize_dirs(rays_d, w.dn, n_rays, st)) return rc;
    if (int rc = nerfart_ray_points(rays_o, w.dn, d_all, R, P, w.pts, nullptr, st)) return rc;
    hipLaunchKernelGGL(k_mid_depths, dim3((unsigned)((Mm + 255) / 256)), dim3(256), 0, st, d_all, R, P, w.d_mid);
    NERFART_HIP(hipGetLastError());
    if (int rc = nerfart_ray_points(rays_o, w.dn, w.d_mid, R, P - 1, w.pts_m, w.view_m, st)) return rc;
    const float *sdf = sdf_state, *nab = nabla_state;
    if (!have_state) {
        if (int rc = nerfart_sdf_nabla_fwd(surf_blob, 1, w.pts, M, 0.f, w.sdf, w.nab, nullptr, w.nabla_ws, w.nabla_ws_bytes, st)) return rc;
        sdf = w.sdf; nab = w.nab;
    }
    if (int rc = nerfart_sdf_nabla_fwd(surf_blob, 1, w.pts_m, Mm, 0.f, w.sdf_m, w.nab_m, w.h7_m, w.nabla_ws, w.nabla_ws_bytes, st)) return rc;
    if (int rc = nerfart_radiance_fwd_dump(rad_blob, view_tiles, w.pts_m, w.view_m, Mm, w.nab_m, w.h7_m, w.rad.rgb, w.rad.dump, st)) return rc;
    if (int rc = nerfart_neus_composite_bwd(n_rays, P, sdf, w.rad.rgb, s, white_bkgd, g_rgb, g_acc, w.g_sdf, w.rad.g_rad, scal + 2, st)) return rc;
    if (int rc = nerfart_radiance_bwd(rad_blob, Mm, w.rad.rgb, w.rad.g_rad, w.rad.dump, w.rad.bdump, w.rad.g_h7, w.rad.g_n, st)) return rc;
    // samples: cotangents of sdf (alpha) and of the nablas (eikonal, neus.py:568-571); no sphere clamp
    if (int rc = nerfart_volsdf_pass2_cotangents(w.pts, sdf, w.g_sdf, nab, nullptr, nullptr, R, P, 0.f, w_eikonal, eik_group_rays, w.sbar, w.nbar, w.eik_ray, st)) return rc;
    if (int rc = add_sums<1>(w.eik_ray, R, w.partial, scal + 3, st)) return rc;
    if (int rc = surf_param_bwd(surf_blob, multires, w.pts, M, w.sbar, nullptr, w.nbar, raw, w.surf, nullptr, st)) return rc;
    // mid-points: cotangents of h7 and of the normal (radiance net)
    const void* a7 = nullptr;
    if (int rc = surf_param_bwd(surf_blob, multires, w.pts_m, Mm, nullptr, w.rad.g_h7, w.rad.g_n, raw, w.surf, &a7, st)) return rc;
    return rad_param_bwd(mv, w.pts_m, w.view_m, w.nab_m, Mm, w.rad.g_rad, a7, train_radiance, raw, w.rad, st);
}

// offsets[2 * 14 + 1]: float offsets of (dW_l, db_l) for the SDF net's layers 0..8, then the radiance net's 0..4, last = total
long long nerfart_folded_grads_layout(int multires, int multires_view, long long* offsets) {
    const int nenc = embed_width(multires), in0r = 3 + embed_width(multires_view) + 3 + 256;
    long long o = 0;
    int k = 0;
    for (int l = 0; l < 9; ++l) {
        const int out = l == 8 ? 257 : (l == 3 ? 256 - nenc : 256), in = l == 0 ? nenc : 256;
        if (offsets) offsets[k] = o;
        ++k; o += (long long)out * in;
        if (offsets) offsets[k] = o;
        ++k; o += out;
    }
    for (int l = 0; l < 5; ++l) {
        const int out = l == 4 ? 3 : 256, in = l == 0 ? in0r : 256;
        if (offsets) offsets[k] = o;
        ++k; o += (long long)out * in;
        if (offsets) offsets[k] = o;
        ++k; o += out;
    }
    if (offsets) offsets[k] = o;
    return o;
}

int nerfart_fold_weight_grads(const float* raw, int multires, int multires_view, float* folded, void* stream) {
    if (!raw || !folded) { set_last_error("fold_weight_grads: null argument"); return 2; }
    if (embed_width(multires) > 64 || 3 + embed_width(multires_view) + 3 > 64) { set_last_error("fold_weight_grads: the narrow operands hold at most 64 columns"); return 2; }
    long long offs[29];
    nerfart_folded_grads_layout(multires, multires_view, offs);
    FoldArgs a;
    a.n_layers = 14;
    a.nenc = embed_width(multires);
    a.nex = 3 + embed_width(multires_view) + 3;
    for (int s = 0; s < N_SECTIONS; ++s) a.sec[s] = section_offset(s);
    long long max_elems = 0;
    for (int l = 0; l < 14; ++l) {
        FoldLayer& L = a.L[l];
        L.w_off = offs[2 * l]; L.b_off = offs[2 * l + 1];
        if (l < 9) {
            L.out = l == 8 ? 257 : (l == 3 ? 256 - a.nenc : 256); L.in = l == 0 ? a.nenc : 256;
            L.kind = l == 0 ? 0 : (l == 4 ? 2 : (l == 8 ? 3 : 1)); L.idx = l;
        } else {
            const int r = l - 9;
            L.out = r == 4 ? 3 : 256; L.in = r == 0 ? a.nex + 256 : 256;
            L.kind = r == 0 ? 4 : (r == 4 ? 6 : 5); L.idx = r;
        }
        const long long n = (long long)L.out * L.in + L.out;
        if (n > max_elems) max_elems = n;
    }
    hipLaunchKernelGGL(k_fold, dim3((unsigned)((max_elems + 255) / 256), 14), dim3(256), 0, (hipStream_t)stream, raw, a, folded);
    NERFART_HIP(hipGetLastError());
    return 0;
}

int nerfart_weight_norm_bwd(const float* dW, const float* weight_v, const float* weight_g, int out_features, int in_features, float* g_weight_v,
                            float* g_weight_g, int accumulate, void* stream) {
    if (out_features <= 0 || in_features <= 0) return 0;
    if (!dW || !weight_v || !weight_g) { set_last_error("weight_norm_bwd: null argument"); return 2; }
    hipLaunchKernelGGL(k_weight_norm_bwd, dim3(out_features), dim3(64), 0, (hipStream_t)stream, dW, weight_v, weight_g, in_features, g_weight_v,
                       g_weight_g, accumulate);
    NERFART_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
