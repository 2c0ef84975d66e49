// mlp_chain.hip - fused positional-encode + per-sample MLP chains on the gfx950 matrix cores.
//
// Replaces (reference, cassiePython/NeRF-Art):
//   K2  sdf_only   : ImplicitSurface.forward            models/base.py:243-263  (+ sphere clamp volsdf.py:341-347)
//   K3a sdf_nabla  : ImplicitSurface.forward_with_nablas models/base.py:265-282  (+ clamp volsdf.py:349-357)
//   K3b radiance   : RadianceNet.forward                 models/base.py:372-391
//
// Design (fp32-exact mode, v_mfma_f32_16x16x4_f32):
//   * out^T[feature][column] = W[feature][k] . h^T[k][column].  A = weight fragment (from LDS),
//     B = activation (a REGISTER of the previous layer's output).  The C layout of the 16x16x4
//     MFMA (lane (g,j), reg r  <->  feature 16t+4g+r of column j) is, up to a fixed permutation of
//     k, the B layout of the next layer, so activations never leave registers between layers; the
//     permutation is folded into the weight packing (nerfart_amd/packing.py).
//   * a workgroup is 8 waves x 16 columns = 128 columns per tile; weights stream L2 -> LDS in
//     "chunks" (two 16-wide k tiles x all 16 output tiles = 32 KiB) with LDS-DMA, double buffered,
//     one barrier per chunk; the weight blob (2.4 MB fp32) stays L2 resident on every XCD.
//   * k-outer order: the 16 accumulator tiles of a layer stay live (64 VGPRs) and, after the
//     activation epilogue, ARE the next layer's input; input tiles die as they are consumed.
//   * K3a evaluates d sdf / d x in forward mode: a column quad = (value, d/dx, d/dy, d/dz) of one
//     point, the activation derivative is broadcast inside the quad with one DPP op.
//   * persistent grid (one workgroup per CU), tiles grid-strided.
#include "mlp_common.h"

namespace nerfart {

constexpr int WG_THREADS = 512;
constexpr int WAVES = 8;
constexpr int XT_MAX = 19;                          // most input tiles of any layer (NeuS radiance layer 0)
constexpr int KT_FLOATS = 16 * 256;                 // one k tile of a chunk: 16 out tiles x (64 lanes x 4)
constexpr int CHUNK_FLOATS_MAX = 2 * KT_FLOATS;     // a chunk holds 1 or 2 k tiles (32 KiB)
constexpr int AUX_FLOATS_MAX = 2560;
constexpr int LDS_FLOATS = 2 * CHUNK_FLOATS_MAX + AUX_FLOATS_MAX + TAB_INTS;   // 76,288 B
// reverse-mode grad(SDF) scratch: softplus'(z_l) as fp32, [layer 8][tile 16][wave 8][lane 64] x 16 B per workgroup
constexpr int D_TILE_STRIDE = WAVES * 1024;
constexpr int D_LAYER_STRIDE = 16 * D_TILE_STRIDE;
constexpr size_t GRADF_WS_PER_WG = 8 * (size_t)D_LAYER_STRIDE;                 // 1 MiB
// The reverse-mode kernel's scratch as a buffer (round 3): descriptor built from kernel arguments and blockIdx (uniform), ONE
// 32-bit voffset register per access (wave * 1024 + lane * 16 + the layer / tile offset, one v_add) - with 64-bit lane pointers
// hipcc hoisted the 128 (layer, tile) addresses out of the tile loop and spilled them (113 VGPR spills, now 3).
// The layer / tile offset is NOT passed as the instruction's scalar soffset: that form (hipcc re-materialises the SGPR between
// back-to-back accesses) returned wrong data on the MI355X for the lanes of waves 4..7 with lane % 16 >= 12, run-dependent
// (tools/archive/dbg_fp32_scratch.py: 12.5 % of the points of every tile, sdf and h7 exact, nabla off by up to 0.17); the same accesses with
// the offset in the vector register, or through plain pointers, are exact.  Cause not established - avoided.
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
struct Scratch {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;                                   // wave * 1024 + lane * 16
};
__device__ __forceinline__ void sc_store(const Scratch& sc, int off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, v), sc.rsrc, sc.voff + off, 0, 0);
}
__device__ __forceinline__ f32x4 sc_load(const Scratch& sc, int off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sc.rsrc, sc.voff + off, 0, 0));
}

using Pipe = PipeT<WAVES, CHUNK_FLOATS_MAX>;

// ---------------------------------------------------------------------------------------
// One dense layer on register-resident activations, in place.
//   in : X[t][r] (t < NT_BASE + nextra) = input feature slot 16t + 4g + r of this lane's column
//   out: X[T][r] (T < 16, or < 14 when !full16) = act(W x + b) feature 16T + 4g + r
// Chunk sequence of a layer (must match packing.py): ceil(NT_BASE/2) chunks of the base k tiles,
// then ceil(nextra/2) chunks of the extra k tiles; every k tile of a chunk is laid out
// [T = 0..15][lane][r] so that lane l reads its A fragments for 4 consecutive k with one
// ds_read_b128.
// What varies between the layers that share a body is uniform run-time state: full16 (false for
// the 217-wide layer: 14 output tiles), nextra (the skip connection / the radiance extras) and,
// for the ReLU family, whether the activation is applied at all.
// TANGENT: columns are quads (value, d/dx, d/dy, d/dz); bias and activation apply to the value
// column, the derivative columns are scaled by act'(z_value).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void mma_ktile(f32x4 (&acc)[16], const f32x4 xt, const float* w, bool full16) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q < 3 || full16) {
            f32x4 a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const f32x4*>(w + (4 * q + i) * 256);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[4 * q + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][r], xt[r], acc[4 * q + i], 0, 0, 0);
        } else {
            f32x4 a[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f32x4*>(w + (12 + i) * 256);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[12 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][r], xt[r], acc[12 + i], 0, 0, 0);
        }
    }
}

template <int NT_BASE, int NT_EXTRA_MAX, bool SOFTPLUS, bool TANGENT, bool DSTORE = false>
__device__ __forceinline__ void run_layer(f32x4 (&X)[XT_MAX], Pipe& p, const float* bias_lds, bool full16, int nextra, bool relu,
                                          const Scratch* sc = nullptr, int sc_layer = 0) {
    const int lane = lane_id();
    const int g = lane >> 4;
    const bool is_val = !TANGENT || ((lane & 3) == 0);
    f32x4 acc[16];
#pragma unroll
    for (int T = 0; T < 16; ++T) acc[T] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < (NT_BASE + 1) / 2; ++c) {
        const float* w = pipe_acquire(p) + lane * 4;
        mma_ktile(acc, X[2 * c], w, full16);
        if (2 * c + 1 < NT_BASE) mma_ktile(acc, X[2 * c + 1], w + KT_FLOATS, full16);
    }
#pragma unroll
    for (int c = 0; c < (NT_EXTRA_MAX + 1) / 2; ++c) {
        if (2 * c < nextra) {
            const float* w = pipe_acquire(p) + lane * 4;
            mma_ktile(acc, X[NT_BASE + 2 * c], w, full16);
            if (2 * c + 1 < NT_EXTRA_MAX) {
                if (2 * c + 1 < nextra) mma_ktile(acc, X[NT_BASE + 2 * c + 1], w + KT_FLOATS, full16);
            }
        }
    }
    // epilogue: bias + activation, results become the next layer's input tiles
#pragma unroll
    for (int T = 0; T < 16; ++T) {
        if (T < 14 || full16) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bias_lds + T * 16 + g * 4);
            f32x4 y, dd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = acc[T][r] + (is_val ? b[r] : 0.f);
                if (SOFTPLUS) {
                    if (TANGENT) {
                        float v, d;
                        softplus100_vd(z, v, d);
                        d = quad_bcast0(d);
                        y[r] = is_val ? v : d * acc[T][r];
                    } else if (DSTORE) {
                        float v, d;
                        softplus100_vd(z, v, d);
                        y[r] = v; dd[r] = d;
                    } else {
                        y[r] = softplus100(z);
                    }
                } else {
                    y[r] = relu ? fmaxf(z, 0.f) : z;
                }
            }
            X[T] = y;
            // reverse-mode kernel, forward sweep: softplus'(z) of this tile to the wave's scratch slot (D_TILE_STRIDE apart)
            if constexpr (DSTORE) sc_store(*sc, sc_layer + T * D_TILE_STRIDE, dd);
        }
    }
}

// sum_k X[k] * row[k] over this lane's slots (row in LDS, natural feature order), then over the
// 4 lane groups: the full 256-long dot product of the column, replicated in all 4 groups.
__device__ __forceinline__ float dot_row16(const f32x4 (&X)[XT_MAX], const float* row_lds, int g) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(row_lds + t * 16 + g * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) s = fmaf(X[t][r], wv[r], s);
    }
    return sum_over_groups(s);
}

// ---------------------------------------------------------------------------------------
// Positional encoding of the SDF net (multires 6 -> 39 features) in "slot" order, 48 slots =
// 3 tiles.  Lane group g < 3 owns coordinate g: [x_g, sin(2^0 x_g), cos(2^0 x_g), ..., sin(2^4 x_g),
// cos(2^4 x_g), 0]; lane group 3 owns the last band: [sin 32x, cos 32x, sin 32y, cos 32y, sin 32z,
// cos 32z, 0 x 6].  (reference Embedder, models/base.py:38-64; slot map in packing.enc_slot_feature)
// q < 0: values.  q = 0..2: derivative of every slot w.r.t. coordinate q.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void encode_slots(float x, float y, float z, int g, int q, f32x4 (&E)[3]) {
    const bool last = (g == 3);
    const float cg = (g == 0) ? x : ((g == 1) ? y : z);
    float a[5], s[5], c[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const float co = (i == 0) ? x : ((i == 1) ? y : z);
        a[i] = last ? ((i < 3) ? co * 32.0f : 0.f) : cg * (float)(1 << i);
        sincosf(a[i], &s[i], &c[i]);
    }
    float m[12];
    if (q < 0) {
#pragma unroll
        for (int k = 0; k < 12; ++k) m[k] = 0.f;
        // coordinate-owner groups
        float mc[12];
        mc[0] = cg;
#pragma unroll
        for (int k = 0; k < 5; ++k) { mc[1 + 2 * k] = s[k]; mc[2 + 2 * k] = c[k]; }
        mc[11] = 0.f;
        float ml[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) ml[k] = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) { ml[2 * i] = s[i]; ml[2 * i + 1] = c[i]; }
#pragma unroll
        for (int k = 0; k < 12; ++k) m[k] = last ? ml[k] : mc[k];
    } else {
        float mc[12], ml[12];
        const bool own = (q == g);
        mc[0] = own ? 1.f : 0.f;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const float f = (float)(1 << k);
            mc[1 + 2 * k] = own ? c[k] * f : 0.f;
            mc[2 + 2 * k] = own ? -(s[k] * f) : 0.f;
        }
        mc[11] = 0.f;
#pragma unroll
        for (int k = 0; k < 12; ++k) ml[k] = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            ml[2 * i] = (q == i) ? c[i] * 32.0f : 0.f;
            ml[2 * i + 1] = (q == i) ? -(s[i] * 32.0f) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) m[k] = last ? ml[k] : mc[k];
    }
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) E[t][r] = m[4 * t + r];
}

// Aux region of a surface blob (floats): bias[l] at l*256 (l = 0..7), final row at 2048, b8[0] at 2304.
constexpr int SURF_AUX_ROW = 2048;
constexpr int SURF_AUX_B8 = 2304;
constexpr int SURF_AUX_FLOATS = 2308;

__device__ __forceinline__ void load_aux(float* aux_lds, const float* blob, const int* hdr, int nfloats) {
    const float* src = blob + hdr[4];
    for (int i = threadIdx.x; i < nfloats; i += WG_THREADS) aux_lds[i] = src[i];
    int* tab = reinterpret_cast<int*>(aux_lds + AUX_FLOATS_MAX);
    if (threadIdx.x < TAB_INTS) tab[threadIdx.x] = hdr[NERFART_HDR_OFFS + threadIdx.x];
    __syncthreads();
}

// The 8 hidden layers of the SDF net, in place on X (out: layer-7 output in X[0..15]).
// Layer 0 (3 input tiles) has its own body; layers 1..7 share one body in a run-time loop.
template <bool TANGENT, bool DSTORE = false>
__device__ __forceinline__ void surface_hidden(f32x4 (&X)[XT_MAX], float px, float py, float pz, int g, int q,
                                               Pipe& p, const float* aux, const Scratch* sc = nullptr) {
    {
        f32x4 E[3];
        encode_slots(px, py, pz, g, q, E);
#pragma unroll
        for (int t = 0; t < 3; ++t) X[t] = E[t];
    }
    run_layer<3, 0, true, TANGENT, DSTORE>(X, p, aux, true, 0, false, sc, 0);
#pragma nounroll
    for (int L = 1; L < 8; ++L) {
        if (L == 4) {
            // skip connection: cat[h(217 -> 14 tiles), enc(3 tiles)] / sqrt(2)   (base.py:248-250)
            // (the encoding is recomputed here rather than kept live through layers 0..3: 12 VGPRs)
            const float rs2 = 1.41421356237309504880f;
            f32x4 E[3];
            encode_slots(px, py, pz, g, q, E);
#pragma unroll
            for (int t = 0; t < 14; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) X[t][r] = X[t][r] / rs2;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) X[14 + t][r] = E[t][r] / rs2;
        }
        // layer 3 has 217 outputs -> 14 tiles (7 zero rows); layer 4 has 17 input tiles
        run_layer<16, 1, true, TANGENT, DSTORE>(X, p, aux + L * 256, L != 3, (L == 4) ? 1 : 0, false, sc, L * D_LAYER_STRIDE);
    }
}

// =======================================================================================
// K2: sdf only.  128 points per workgroup tile, 16 per wave.
// =======================================================================================
__global__ void __launch_bounds__(WG_THREADS, 2)
k_sdf_only(const float* __restrict__ blob, PointSrc src, float R_bg, float* __restrict__ sdf_out, int out_stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int* hdr = reinterpret_cast<const int*>(blob);
    float* aux = smem + 2 * CHUNK_FLOATS_MAX;
    const int lane = lane_id(), g = lane >> 4, j = lane & 15, wv = wave_id();
    load_aux(aux, blob, hdr, SURF_AUX_FLOATS);

    const unsigned ntiles = (src.M + 127u) / 128u;
    if (blockIdx.x >= ntiles) return;
    Pipe p{blob, reinterpret_cast<const int*>(aux + AUX_FLOATS_MAX), smem, hdr[2], 0, 0, 0, 0, false};
    p.wrap = (blockIdx.x + gridDim.x) < ntiles;
    pipe_start(p);

    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        p.wrap = (tile + gridDim.x) < ntiles;
        const unsigned m = tile * 128u + wv * 16 + j;
        const Pt pt = fetch_point(src, m, false);
        f32x4 X[XT_MAX];
        surface_hidden<false>(X, pt.x, pt.y, pt.z, g, -1, p, aux);
        float sdf = dot_row16(X, aux + SURF_AUX_ROW, g) + aux[SURF_AUX_B8];
        if (R_bg > 0.f) sdf = fminf(sdf, R_bg - sqrtf(pt.x * pt.x + pt.y * pt.y + pt.z * pt.z));
        if (g == 0 && m < src.M) {
            if (src.pts) sdf_out[m] = sdf;
            else {
                const unsigned slot = m / (unsigned)src.n_per_ray;
                sdf_out[(size_t)slot * out_stride + (m - slot * (unsigned)src.n_per_ray)] = sdf;
            }
        }
    }
}

// =======================================================================================
// K3a: sdf + nabla (+ last hidden activation h7 for the radiance kernel).
// 32 points per workgroup tile: 4 points x (value, d/dx, d/dy, d/dz) per wave.
// =======================================================================================
__global__ void __launch_bounds__(WG_THREADS, 2)
k_sdf_nabla(const float* __restrict__ blob, PointSrc src, float R_bg, float* __restrict__ sdf_out,
            float* __restrict__ nabla_out, float* __restrict__ h7_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int* hdr = reinterpret_cast<const int*>(blob);
    float* aux = smem + 2 * CHUNK_FLOATS_MAX;
    const int lane = lane_id(), g = lane >> 4, j = lane & 15, wv = wave_id();
    const int cq = j & 3;     // 0 = value column, 1..3 = derivative w.r.t. x, y, z
    load_aux(aux, blob, hdr, SURF_AUX_FLOATS);

    const unsigned ntiles = (src.M + 31u) / 32u;
    if (blockIdx.x >= ntiles) return;
    Pipe p{blob, reinterpret_cast<const int*>(aux + AUX_FLOATS_MAX), smem, hdr[2], 0, 0, 0, 0, false};
    p.wrap = (blockIdx.x + gridDim.x) < ntiles;
    pipe_start(p);

    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        p.wrap = (tile + gridDim.x) < ntiles;
        const unsigned m = tile * 32u + wv * 4 + (j >> 2);
        const Pt pt = fetch_point(src, m, false);
        f32x4 X[XT_MAX];
        surface_hidden<true>(X, pt.x, pt.y, pt.z, g, cq - 1, p, aux);
        float v = dot_row16(X, aux + SURF_AUX_ROW, g);
        if (m < src.M) {
            if (cq == 0) {
                float sdf = v + aux[SURF_AUX_B8];
                if (R_bg > 0.f) {                       // sdf[d_bg < sdf] = d_bg, nabla untouched (volsdf.py:351-356)
                    const float d_bg = R_bg - sqrtf(pt.x * pt.x + pt.y * pt.y + pt.z * pt.z);
                    sdf = (d_bg < sdf) ? d_bg : sdf;
                }
                if (g == 0) sdf_out[m] = sdf;
                if (h7_out) {
                    float* dst = h7_out + (size_t)m * 256 + g * 4;
#pragma unroll
                    for (int t = 0; t < 16; ++t) *reinterpret_cast<f32x4*>(dst + t * 16) = X[t];
                }
            } else if (g == 0) {
                nabla_out[(size_t)m * 3 + (cq - 1)] = v;
            }
        }
    }
}

// =======================================================================================
// K3a, reverse mode (what nerfart_sdf_nabla_fwd runs at precision 0): sdf + nabla + h7 with ONE column per point, 128 points
// per workgroup tile.  Forward sweep = K2 with softplus'(z_l) of every layer parked in the wave's scratch slot (fp32: this is the
// exact-product mode); then d sdf / d a_{l-1} = W_l^T (d sdf / d a_l . softplus'(z_l)), l = 7 .. 0, through the transposed
// chunks that follow the forward program in the blob (packing.surface_plan: header word 6): 2 chain sweeps per point instead of
// the 4 columns (value + 3 tangents) of k_sdf_nabla.  The 48 encoding slots of layer 4's input (skip connection) and of layer 0
// come out of 3-tile "tails" and meet the encoding's Jacobian in registers.  (reference: autograd.grad of sdf w.r.t. x,
// models/base.py:252-263)
// =======================================================================================
// g <- (W^T g) . softplus'(z of the layer below) * scale:  NT_K k tiles (the layer's outputs), 16 or 14 output tiles
template <int NT_K>
__device__ __forceinline__ void run_layer_T(f32x4 (&X)[XT_MAX], Pipe& p, bool full16, const Scratch& sc, int layer_below, float scale) {
    const int lane = lane_id();
    f32x4 acc[16];
#pragma unroll
    for (int T = 0; T < 16; ++T) acc[T] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NT_K / 2; ++c) {
        const float* w = pipe_acquire(p) + lane * 4;
        mma_ktile(acc, X[2 * c], w, full16);
        mma_ktile(acc, X[2 * c + 1], w + KT_FLOATS, full16);
    }
    // X is dead from here: its registers take the softplus' tiles (all loads in flight together), then the products
#pragma unroll
    for (int T = 0; T < 16; ++T)
        if (T < 14 || full16) X[T] = sc_load(sc, layer_below * D_LAYER_STRIDE + T * D_TILE_STRIDE);
#pragma unroll
    for (int T = 0; T < 16; ++T)
        if (T < 14 || full16) {
#pragma unroll
            for (int r = 0; r < 4; ++r) X[T][r] = acc[T][r] * scale * X[T][r];
        }
}
// E += (W_enc^T g) * scale: the 48 encoding slots (3 output tiles), 16 k tiles in 2 chunks of 8
__device__ __forceinline__ void run_tail_T(const f32x4 (&X)[XT_MAX], Pipe& p, f32x4 (&E)[3], float scale) {
    const int lane = lane_id();
    f32x4 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float* w = pipe_acquire(p) + lane * 4;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            f32x4 a[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) a[i] = *reinterpret_cast<const f32x4*>(w + kt * 768 + i * 256);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 3; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][r], X[8 * c + kt][r], acc[i], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) E[i][r] += acc[i][r] * scale;
}

__global__ void __launch_bounds__(WG_THREADS, 2)
k_sdf_grad(const float* __restrict__ blob, PointSrc src, float R_bg, float* __restrict__ sdf_out,
           float* __restrict__ nabla_out, float* __restrict__ h7_out, char* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int* hdr = reinterpret_cast<const int*>(blob);
    float* aux = smem + 2 * CHUNK_FLOATS_MAX;
    const int lane = lane_id(), g = lane >> 4, j = lane & 15, wv = wave_id();
    load_aux(aux, blob, hdr, SURF_AUX_FLOATS);

    const unsigned ntiles = (src.M + 127u) / 128u;
    if (blockIdx.x >= ntiles) return;
    Pipe p{blob, reinterpret_cast<const int*>(aux + AUX_FLOATS_MAX), smem, hdr[6], 0, 0, 0, 0, false};     // forward + reverse chunks
    p.wrap = (blockIdx.x + gridDim.x) < ntiles;
    pipe_start(p);
    Scratch sc;
    sc.rsrc = __builtin_amdgcn_make_buffer_rsrc(ws + (size_t)blockIdx.x * GRADF_WS_PER_WG, 0, (int)GRADF_WS_PER_WG, 0x00020000);
    sc.voff = wv * 1024 + lane * 16;
    const float rs2 = 0.70710678118654752440f;

    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        p.wrap = (tile + gridDim.x) < ntiles;
        const unsigned m = tile * 128u + wv * 16 + j;
        const Pt pt = fetch_point(src, m, false);
        f32x4 X[XT_MAX];
        surface_hidden<false, true>(X, pt.x, pt.y, pt.z, g, -1, p, aux, &sc);
        float sdf = dot_row16(X, aux + SURF_AUX_ROW, g) + aux[SURF_AUX_B8];
        if (m < src.M) {
            if (R_bg > 0.f) {                           // sdf[d_bg < sdf] = d_bg, nabla untouched (volsdf.py:351-356)
                const float d_bg = R_bg - sqrtf(pt.x * pt.x + pt.y * pt.y + pt.z * pt.z);
                sdf = (d_bg < sdf) ? d_bg : sdf;
            }
            if (g == 0) sdf_out[m] = sdf;
            if (h7_out) {
                float* dst = h7_out + (size_t)m * 256 + g * 4;
#pragma unroll
                for (int t = 0; t < 16; ++t) *reinterpret_cast<f32x4*>(dst + t * 16) = X[t];
            }
        }
        // d sdf / d z_7 = row . softplus'(z_7)
#pragma unroll
        for (int T = 0; T < 16; ++T) {
            const f32x4 d = sc_load(sc, 7 * D_LAYER_STRIDE + T * D_TILE_STRIDE);
            const f32x4 row = *reinterpret_cast<const f32x4*>(aux + SURF_AUX_ROW + T * 16 + g * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) X[T][r] = row[r] * d[r];
        }
        f32x4 E[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) E[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma nounroll
        for (int L = 7; L >= 5; --L) run_layer_T<16>(X, p, true, sc, L - 1, 1.0f);
        run_tail_T(X, p, E, rs2);                                                           // layer 4, encoding columns
        run_layer_T<16>(X, p, false, sc, 3, rs2);                        // layer 4, the 217 hidden columns
        run_layer_T<14>(X, p, true, sc, 2, 1.0f);                        // layer 3 (217 outputs = 14 k tiles)
#pragma nounroll
        for (int L = 2; L >= 1; --L) run_layer_T<16>(X, p, true, sc, L - 1, 1.0f);
        run_tail_T(X, p, E, 1.0f);                                                          // layer 0
        if (nabla_out) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                f32x4 J[3];
                encode_slots(pt.x, pt.y, pt.z, g, q, J);
                float a = 0.f;
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) a = fmaf(E[t][r], J[t][r], a);
                a = sum_over_groups(a);
                if (g == 0 && m < src.M) nabla_out[(size_t)m * 3 + q] = a;
            }
        }
    }
}

// =======================================================================================
// K3b: radiance net.  Input per point: x, view dir, nabla (raw), h7 (layer-7 activation of the SDF
// net; the geometry feature = W8[1:] h7 + b8[1:] is this kernel's first, activation-free layer).
// VE = number of extra input tiles: 1 (VolSDF: [x, v, n] = 9 slots) or 3 (NeuS: [x, embed4(v), n] = 33).
// Aux (floats): bias of layers A, R0..R3 at l*256; final 3 rows at 1280 + 256*c; final bias at 2048.
// =======================================================================================
constexpr int RAD_AUX_ROWS = 1280;
constexpr int RAD_AUX_BF = 2048;
constexpr int RAD_AUX_FLOATS = 2052;

template <int VE>
__device__ __forceinline__ void radiance_extras(const Pt& pt, float nx, float ny, float nz, int g, f32x4 (&X)[XT_MAX]) {
    constexpr int NE = (VE == 1) ? 9 : 33;
    float ex[VE * 16];
#pragma unroll
    for (int k = 0; k < VE * 16; ++k) ex[k] = 0.f;
    ex[0] = pt.x; ex[1] = pt.y; ex[2] = pt.z;
    if (VE == 1) {
        ex[3] = pt.vx; ex[4] = pt.vy; ex[5] = pt.vz;
    } else {
        const float v[3] = {pt.vx, pt.vy, pt.vz};
#pragma unroll
        for (int c = 0; c < 3; ++c) ex[3 + c] = v[c];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float s, co;
                sincosf(v[c] * (float)(1 << k), &s, &co);
                ex[6 + 6 * k + c] = s;
                ex[6 + 6 * k + 3 + c] = co;
            }
    }
    ex[NE - 3] = nx; ex[NE - 2] = ny; ex[NE - 1] = nz;
#pragma unroll
    for (int t = 0; t < VE; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e0 = ex[16 * t + r], e1 = ex[16 * t + 4 + r], e2 = ex[16 * t + 8 + r], e3 = ex[16 * t + 12 + r];
            X[16 + t][r] = (g == 0) ? e0 : ((g == 1) ? e1 : ((g == 2) ? e2 : e3));
        }
}

template <int VE>
__global__ void __launch_bounds__(WG_THREADS, 2)
k_radiance(const float* __restrict__ blob, PointSrc src, const float* __restrict__ nabla_in,
           const float* __restrict__ h7_in, float* __restrict__ rgb_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int* hdr = reinterpret_cast<const int*>(blob);
    float* aux = smem + 2 * CHUNK_FLOATS_MAX;
    const int lane = lane_id(), g = lane >> 4, j = lane & 15, wv = wave_id();
    load_aux(aux, blob, hdr, RAD_AUX_FLOATS);

    const unsigned ntiles = (src.M + 127u) / 128u;
    if (blockIdx.x >= ntiles) return;
    Pipe p{blob, reinterpret_cast<const int*>(aux + AUX_FLOATS_MAX), smem, hdr[2], 0, 0, 0, 0, false};
    p.wrap = (blockIdx.x + gridDim.x) < ntiles;
    pipe_start(p);

    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        p.wrap = (tile + gridDim.x) < ntiles;
        const unsigned m = tile * 128u + wv * 16 + j;
        const bool valid = m < src.M;
        const Pt pt = fetch_point(src, m, true);
        f32x4 X[XT_MAX];
        float nx = 0.f, ny = 0.f, nz = 0.f;
        if (valid) {
            nx = nabla_in[(size_t)m * 3 + 0]; ny = nabla_in[(size_t)m * 3 + 1]; nz = nabla_in[(size_t)m * 3 + 2];
            const float* hsrc = h7_in + (size_t)m * 256 + g * 4;
#pragma unroll
            for (int t = 0; t < 16; ++t) X[t] = *reinterpret_cast<const f32x4*>(hsrc + t * 16);
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) X[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        radiance_extras<VE>(pt, nx, ny, nz, g, X);
        // L = 0: geometry feature = W8[1:257] h7 + b8[1:257], no activation (base.py:253-256);
        // L = 1: [feat, x, v, n] -> 256 ReLU; L = 2..4: 256 -> 256 ReLU.
#pragma nounroll
        for (int L = 0; L < 5; ++L)
            run_layer<16, VE, false, false>(X, p, aux + L * 256, true, (L == 1) ? VE : 0, L != 0);
        const float c0 = sigmoidf_(dot_row16(X, aux + RAD_AUX_ROWS + 0, g) + aux[RAD_AUX_BF + 0]);
        const float c1 = sigmoidf_(dot_row16(X, aux + RAD_AUX_ROWS + 256, g) + aux[RAD_AUX_BF + 1]);
        const float c2 = sigmoidf_(dot_row16(X, aux + RAD_AUX_ROWS + 512, g) + aux[RAD_AUX_BF + 2]);
        if (valid && g < 3) rgb_out[(size_t)m * 3 + g] = (g == 0) ? c0 : ((g == 1) ? c1 : c2);
    }
}

template <typename K, typename... Args>
static int launch_chain(int prof_cls, long long units, K kernel, unsigned ntiles, hipStream_t stream, Args... args) {
    const size_t lds = LDS_FLOATS * sizeof(float);
    NERFART_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = ntiles < (unsigned)num_cus() ? ntiles : (unsigned)num_cus();
    void* ph = nullptr;
    if (profile_enabled()) profile_open(prof_cls, units, stream, &ph);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(WG_THREADS), lds, stream, args...);
    profile_close(ph, stream);
    NERFART_HIP(hipGetLastError());
    return 0;
}

}  // namespace nerfart

using namespace nerfart;

namespace nerfart {
int sdf_bf16(const float* blob, const PointSrc& s, float R_bg, float* out, int out_stride, hipStream_t st);
int sdf_nabla_bf16(const float* blob, const PointSrc& s, float R_bg, float* sdf, float* nabla, float* h7, hipStream_t st);
int radiance_bf16(const float* blob, int view_tiles, const PointSrc& s, const float* nabla, const float* h7, float* rgb, hipStream_t st);
size_t sdf_grad_ws_bytes();
size_t radiance_dump_bytes(long long M);
size_t sdf_fwd2_dump_bytes(long long M);
size_t sdf_bwd2_dump_bytes(long long M);
int sdf_fwd2_bf16(const float* blob, long long M, const float* pts, const float* dirv, void* dump, hipStream_t st);
int sdf_bwd2_bf16(const float* blob, long long M, const float* gbar_h7, const float* gbar_sdf, void* f2_dump, void* r2_dump, hipStream_t st);
int radiance_fwd_dump_bf16(const float* blob, int view_tiles, const PointSrc& s, const float* nabla, const float* h7, float* rgb, void* dump, hipStream_t st);
int radiance_bwd_bf16(const float* blob, long long M, const float* rgb, const float* g_rgb, void* fwd_dump, void* bwd_dump, float* g_h7,
                      float* g_n, hipStream_t st);
int sdf_grad_bf16(const float* blob, const PointSrc& s, float R_bg, float* sdf, float* nabla, float* h7, void* ws, hipStream_t st);
// precision 4 (csrc/mlp_chain_f16x2.hip: the same three kernels with the 2-MFMA fp16 split)
int sdf_f16x2(const float* blob, const PointSrc& s, float R_bg, float* out, int out_stride, hipStream_t st);
int radiance_f16x2(const float* blob, int view_tiles, const PointSrc& s, const float* nabla, const float* h7, float* rgb, hipStream_t st);
int sdf_grad_f16x2(const float* blob, const PointSrc& s, float R_bg, float* sdf, float* nabla, float* h7, void* ws, hipStream_t st);
// precision 5 (csrc/mlp_chain_f16x1.hip: K2 alone with ONE MFMA per product, on the precision-4 blob; the SDF-only entry points, i.e. the sampler's queries)
int sdf_f16x1(const float* blob, const PointSrc& s, float R_bg, float* out, int out_stride, hipStream_t st);
// bytes of the reverse-mode kernels' softplus' scratch (one private region per resident workgroup): caller owned
static size_t nabla_ws_bytes(int precision) {
    if (precision == 0) return (size_t)num_cus() * GRADF_WS_PER_WG;
    if (precision == 1 || precision == 4) return sdf_grad_ws_bytes();
    return 0;                                         // forward-mode tangent kernels (2, 3): none
}
static int check_precision(int precision, bool allow_fwd_tangents = false, const void* blob = nullptr, const char* who = "point query", bool sdf_only = false) {
    if (precision == 5 && !sdf_only) {
        set_last_error("precision 5 (1-MFMA 'fp16x1') exists for the SDF-only queries of Algorithm 1's sampler (nerfart_sdf_fwd / nerfart_sdf_fwd_rays / the sampler "
                       "stage of the renderers): no value that reaches a pixel is computed in it");
        return 2;
    }
    if (precision == 5 || precision == 0 || precision == 1 || precision == 4 || (allow_fwd_tangents && (precision == 2 || precision == 3)))
        return blob ? blob_term_check(blob, term_of_precision(precision), who) : 0;
    set_last_error("precision must be 0 (fp32-exact MFMA), 1 (split-bf16 'bf16x3' MFMA) or 4 (2-MFMA 'fp16x2', measurement variant)");
    return 2;
}
// precision 0: reverse-mode kernel; precision 3: the forward-mode tangent quads of k_sdf_nabla (kept for cross-checks - same
// blob, 2x the matrix work)
static int sdf_nabla_f32_dispatch(int precision, const float* blob, const PointSrc& s, float R_bg, float* sdf, float* nabla, float* h7,
                                  void* ws, hipStream_t st) {
    const long long M = s.M;
    if (precision == 3) return launch_chain(1, M, k_sdf_nabla, (unsigned)((M + 31) / 32), st, blob, s, R_bg, sdf, nabla, h7);
    const unsigned ntiles = (unsigned)((M + 127) / 128);
    return launch_chain(1, M, k_sdf_grad, ntiles, st, blob, s, R_bg, sdf, nabla, h7, (char*)ws);   // grid <= one workgroup per CU
}
// precision 1: reverse-mode kernel (one column per point); precision 2: the forward-mode tangent quads (kept for
// cross-checks - same blob, 2.1x the matrix work)
static int sdf_nabla_bf16_dispatch(int precision, const float* blob, const PointSrc& s, float R_bg, float* sdf, float* nabla,
                                   float* h7, void* ws, hipStream_t st) {
    if (precision == 2) return sdf_nabla_bf16(blob, s, R_bg, sdf, nabla, h7, st);
    if (precision == 4) return sdf_grad_f16x2(blob, s, R_bg, sdf, nabla, h7, ws, st);
    return sdf_grad_bf16(blob, s, R_bg, sdf, nabla, h7, ws, st);
}
static int check_nabla_ws(int precision, const void* ws, long long ws_bytes) {
    const size_t need = nabla_ws_bytes(precision);
    if (need && (!ws || ws_bytes < (long long)need)) {
        set_last_error("sdf_nabla_fwd: workspace missing or smaller than nerfart_sdf_nabla_workspace_bytes(precision)");
        return 2;
    }
    return 0;
}
}  // namespace nerfart

extern "C" {

int nerfart_sdf_fwd(const float* blob, int precision, const float* pts, long long M, float R_bg, float* sdf_out, void* stream) {
    if (int rc = check_precision(precision, false, blob, "nerfart_sdf_fwd", true)) return rc;
    if (int rc = check_M(M)) return rc;
    if (M == 0) return 0;
    PointSrc s = make_src(pts, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0, M);
    if (int rc = validate_src(s)) return rc;
    if (precision == 5) return sdf_f16x1(blob, s, R_bg, sdf_out, 0, (hipStream_t)stream);
    if (precision == 4) return sdf_f16x2(blob, s, R_bg, sdf_out, 0, (hipStream_t)stream);
    if (precision == 1) return sdf_bf16(blob, s, R_bg, sdf_out, 0, (hipStream_t)stream);
    return launch_chain(0, M, k_sdf_only, (unsigned)((M + 127) / 128), (hipStream_t)stream, blob, s, R_bg, sdf_out, 0);
}

int nerfart_sdf_fwd_rays(const float* blob, int precision, const float* rays_o, const float* rays_d, const int* ray_idx,
                         const float* depth, int n_slots, int n_per_ray, int depth_stride, float R_bg,
                         float* sdf_out, int out_stride, void* stream) {
    const long long M = (long long)n_slots * n_per_ray;
    if (int rc = check_M(M)) return rc;
    if (M == 0) return 0;
    PointSrc s = make_src(nullptr, nullptr, rays_o, rays_d, ray_idx, depth, n_per_ray, depth_stride, M);
    if (int rc = validate_src(s)) return rc;
    if (int rc = check_precision(precision, false, blob, "nerfart_sdf_fwd_rays", true)) return rc;
    if (precision == 5) return sdf_f16x1(blob, s, R_bg, sdf_out, out_stride, (hipStream_t)stream);
    if (precision == 4) return sdf_f16x2(blob, s, R_bg, sdf_out, out_stride, (hipStream_t)stream);
    if (precision == 1) return sdf_bf16(blob, s, R_bg, sdf_out, out_stride, (hipStream_t)stream);
    return launch_chain(0, M, k_sdf_only, (unsigned)((M + 127) / 128), (hipStream_t)stream, blob, s, R_bg, sdf_out, out_stride);
}

long long nerfart_sdf_nabla_workspace_bytes(int precision) { return (long long)nabla_ws_bytes(precision); }

int nerfart_sdf_nabla_fwd(const float* blob, int precision, const float* pts, long long M, float R_bg, float* sdf_out,
                          float* nabla_out, float* h7_out, void* workspace, long long workspace_bytes, void* stream) {
    if (int rc = check_M(M)) return rc;
    if (M == 0) return 0;
    PointSrc s = make_src(pts, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0, M);
    if (int rc = validate_src(s)) return rc;
    if (int rc = check_precision(precision, true, blob, "nerfart_sdf_nabla_fwd")) return rc;
    if (int rc = check_nabla_ws(precision, workspace, workspace_bytes)) return rc;
    if (precision == 1 || precision == 2 || precision == 4) return sdf_nabla_bf16_dispatch(precision, blob, s, R_bg, sdf_out, nabla_out, h7_out, workspace, (hipStream_t)stream);
    return sdf_nabla_f32_dispatch(precision, blob, s, R_bg, sdf_out, nabla_out, h7_out, workspace, (hipStream_t)stream);
}

int nerfart_sdf_nabla_fwd_rays(const float* blob, int precision, const float* rays_o, const float* rays_d, const int* ray_idx,
                               const float* depth, int n_slots, int n_per_ray, int depth_stride, float R_bg,
                               float* sdf_out, float* nabla_out, float* h7_out, void* workspace, long long workspace_bytes, void* stream) {
    const long long M = (long long)n_slots * n_per_ray;
    if (int rc = check_M(M)) return rc;
    if (M == 0) return 0;
    PointSrc s = make_src(nullptr, nullptr, rays_o, rays_d, ray_idx, depth, n_per_ray, depth_stride, M);
    if (int rc = validate_src(s)) return rc;
    if (int rc = check_precision(precision, true, blob, "nerfart_sdf_nabla_fwd_rays")) return rc;
    if (int rc = check_nabla_ws(precision, workspace, workspace_bytes)) return rc;
    if (precision == 1 || precision == 2 || precision == 4) return sdf_nabla_bf16_dispatch(precision, blob, s, R_bg, sdf_out, nabla_out, h7_out, workspace, (hipStream_t)stream);
    return sdf_nabla_f32_dispatch(precision, blob, s, R_bg, sdf_out, nabla_out, h7_out, workspace, (hipStream_t)stream);
}

int nerfart_radiance_fwd(const float* blob, int precision, int view_tiles, const float* pts, const float* view, long long M,
                         const float* nabla, const float* h7, float* rgb_out, void* stream) {
    if (int rc = check_M(M)) return rc;
    if (M == 0) return 0;
    if (!view) { set_last_error("radiance_fwd: view dirs required in pts mode"); return 2; }
    PointSrc s = make_src(pts, view, nullptr, nullptr, nullptr, nullptr, 1, 0, M);
    if (int rc = validate_src(s)) return rc;
    if (int rc = check_precision(precision, false, blob, "nerfart_radiance_fwd")) return rc;
    if (precision == 4) return radiance_f16x2(blob, view_tiles, s, nabla, h7, rgb_out, (hipStream_t)stream);
    if (precision == 1) return radiance_bf16(blob, view_tiles, s, nabla, h7, rgb_out, (hipStream_t)stream);
    const unsigned nt = (unsigned)((M + 127) / 128);
    if (view_tiles == 1) return launch_chain(2, M, k_radiance<1>, nt, (hipStream_t)stream, blob, s, nabla, h7, rgb_out);
    if (view_tiles == 3) return launch_chain(2, M, k_radiance<3>, nt, (hipStream_t)stream, blob, s, nabla, h7, rgb_out);
    set_last_error("radiance_fwd: view_tiles must be 1 (raw view dirs) or 3 (multires_view = 4)");
    return 2;
}

int nerfart_radiance_fwd_rays(const float* blob, int precision, int view_tiles, const float* rays_o, const float* rays_d,
                              const int* ray_idx, const float* depth, int n_slots, int n_per_ray, int depth_stride,
                              const float* nabla, const float* h7, float* rgb_out, void* stream) {
    const long long M = (long long)n_slots * n_per_ray;
    if (int rc = check_M(M)) return rc;
    if (M == 0) return 0;
    PointSrc s = make_src(nullptr, nullptr, rays_o, rays_d, ray_idx, depth, n_per_ray, depth_stride, M);
    if (int rc = validate_src(s)) return rc;
    if (int rc = check_precision(precision, false, blob, "nerfart_radiance_fwd_rays")) return rc;
    if (precision == 4) return radiance_f16x2(blob, view_tiles, s, nabla, h7, rgb_out, (hipStream_t)stream);
    if (precision == 1) return radiance_bf16(blob, view_tiles, s, nabla, h7, rgb_out, (hipStream_t)stream);
    const unsigned nt = (unsigned)((M + 127) / 128);
    if (view_tiles == 1) return launch_chain(2, M, k_radiance<1>, nt, (hipStream_t)stream, blob, s, nabla, h7, rgb_out);
    if (view_tiles == 3) return launch_chain(2, M, k_radiance<3>, nt, (hipStream_t)stream, blob, s, nabla, h7, rgb_out);
    set_last_error("radiance_fwd: view_tiles must be 1 (raw view dirs) or 3 (multires_view = 4)");
    return 2;
}


// ---- radiance net with activation dumps + its backward (row a19; split-bf16 blobs only) ----------------------
long long nerfart_radiance_dump_bytes(long long M) { return (long long)radiance_dump_bytes(M); }

int nerfart_radiance_fwd_dump(const float* rad_blob, int view_tiles, const float* pts, const float* view, long long M,
                              const float* nabla, const float* h7, float* rgb_out, void* dump, void* stream) {
    if (int rc = check_M(M)) return rc;
    if (M == 0) return 0;
    PointSrc s = make_src(pts, view, nullptr, nullptr, nullptr, nullptr, 1, 0, M);
    if (int rc = validate_src(s)) return rc;
    if (!dump) { set_last_error("radiance_fwd_dump: dump buffer is NULL"); return 2; }
    return radiance_fwd_dump_bf16(rad_blob, view_tiles, s, nabla, h7, rgb_out, dump, (hipStream_t)stream);
}

int nerfart_radiance_bwd(const float* rad_blob, long long M, const float* rgb, const float* g_rgb, void* fwd_dump, void* bwd_dump,
                         float* g_h7_out, float* g_n_out, void* stream) {
    if (int rc = check_M(M)) return rc;
    if (M == 0) return 0;
    if (!fwd_dump || !bwd_dump) { set_last_error("radiance_bwd: dump buffers are NULL"); return 2; }
    return radiance_bwd_bf16(rad_blob, M, rgb, g_rgb, fwd_dump, bwd_dump, g_h7_out, g_n_out, (hipStream_t)stream);
}

// ---- second-order backward of the SDF net (row a19; split-bf16 surface blob) -------------------------------------
long long nerfart_sdf_fwd2_dump_bytes(long long M) { return (long long)sdf_fwd2_dump_bytes(M); }
long long nerfart_sdf_bwd2_dump_bytes(long long M) { return (long long)sdf_bwd2_dump_bytes(M); }

int nerfart_sdf_fwd2(const float* surf_blob, const float* pts, const float* dir, long long M, void* f2_dump, void* stream) {
    if (int rc = check_M(M)) return rc;
    if (M == 0) return 0;
    if (!pts || !dir || !f2_dump) { set_last_error("sdf_fwd2: NULL argument"); return 2; }
    return sdf_fwd2_bf16(surf_blob, M, pts, dir, f2_dump, (hipStream_t)stream);
}

int nerfart_sdf_bwd2(const float* surf_blob, long long M, const float* gbar_h7, const float* gbar_sdf, void* f2_dump, void* r2_dump,
                     void* stream) {
    if (int rc = check_M(M)) return rc;
    if (M == 0) return 0;
    if (!gbar_h7 || !gbar_sdf || !f2_dump || !r2_dump) { set_last_error("sdf_bwd2: NULL argument"); return 2; }
    return sdf_bwd2_bf16(surf_blob, M, gbar_h7, gbar_sdf, f2_dump, r2_dump, (hipStream_t)stream);
}
}  // extern "C"
