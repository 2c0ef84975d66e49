// mlp_chain_bf16.hip - the chained-MLP kernels on the bf16 matrix cores with SPLIT operands
// ("bf16x3"): every fp32 operand is split x = hi + lo (hi = bf16_rne(x), lo = bf16_rne(x - hi)) and
// a.b is evaluated as a_hi.b_hi + a_hi.b_lo + a_lo.b_hi with fp32 accumulation - three
// v_mfma_f32_16x16x32_bf16 per 32-deep k-step.  The dropped a_lo.b_lo term and the residual of the
// two-term split are each ~2^-16 relative per product (measured on the SDF net: 2e-5 max / 3e-6
// mean absolute sdf error vs fp64, against 2.5e-6 for plain fp32), i.e. ~17 significand bits where
// TF32 (what the reference's published RTX 3090 number used) has 11.  The bf16 MFMA rate is 16x the
// fp32 MFMA rate, so the split form is 16/3 = 5.3x the fp32-exact kernels of mlp_chain.hip.
//
// Structure: 8 waves x 16 columns (points) per workgroup, two waves per SIMD; activations never leave
// registers; weights stream L2 -> LDS by LDS-DMA in 64 KiB chunks (2 k-steps x 16 output tiles x (hi, lo)
// fragments), double buffered, one barrier per chunk.
//   * 16x16x32 layouts: lane (g = lane>>4, j = lane&15).  A: lane (g,i) holds 8 k-slots of row i; B: lane
//     (g,j) holds 8 k-slots of column j; C: reg r of tile T <-> row 16T + 4g + r.  Which feature a k-slot
//     means is our choice (the MFMA only pairs slot (g,e) of A with slot (g,e) of B): the 4+4 registers of
//     output tiles 2u and 2u+1 become, after activation + split + v_cvt_pk_bf16_f32, "unit" u (32 slots) of
//     the next layer's B operand directly in registers; packing.py (bf16 plans) applies the permutation.
//   * SOFTWARE PIPELINE ACROSS LAYERS.  All 8 waves of a workgroup meet at the chunk barrier, so the two
//     waves of a SIMD are in the same phase: an epilogue (softplus + split: ~10 VALU issues and 2
//     transcendentals per value) executed between layers leaves the matrix cores idle in both.  Instead a
//     layer keeps TWO accumulator sets: Q (being accumulated) and P (the previous layer's pre-activations).
//     k-step u of the layer needs only input unit u = act(P tiles 2u, 2u+1), which is computed in slices
//     placed between the MFMA triples of k-step u-1 (unit 0: in the last k-step of the previous layer, whose
//     tiles 0 and 1 are final by then).  Every MFMA triple is followed by ~5 independent VALU instructions
//     and (first 4 triples of a k-step) one 1 KiB LDS-DMA piece of the next chunk, so VALU, DMA issue and
//     matrix work overlap inside each wave and across the two waves of the SIMD.
//   * A fragments are read from LDS two triples ahead with inline-asm ds_read_b128 and counted lgkmcnt waits
//     (hipcc sinks plain reads back to their use and waits lgkmcnt(0) every 3 MFMAs).
// History (git): 32x32x16, one wave per SIMD: 42 % MFMA utilisation; 16x16x32 with the epilogue between
// layers: 51 %; ablations attributed the rest to the serialised epilogue (25 %) and LDS-DMA issue (18 %).
#include "mlp_bf16_core.h"

namespace nerfart {
namespace b16 {

// =======================================================================================
// K2 (split bf16): sdf only, 128 points per workgroup tile, 16 per wave.
// =======================================================================================
__global__ void __launch_bounds__(WG_THREADS, 2)
k_sdf_only_bf16(const float* __restrict__ blob, PointSrc src, float R_bg, float* __restrict__ sdf_out, int out_stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int* hdr = reinterpret_cast<const int*>(blob);
    float* aux = smem + 2 * CHUNK_FLOATS;
    const int lane = lane_id(), g = lane >> 4, j = lane & 15, wv = wave_id();
    load_aux(aux, blob, hdr, SURF_AUX_FLOATS);
    const unsigned ntiles = (src.M + 127u) / 128u;
    if (blockIdx.x >= ntiles) return;
    Stream s = make_stream(blob, aux, smem, hdr[2]);
    s.wrap = (blockIdx.x + gridDim.x) < ntiles;
    stream_start(s);
#ifdef NERFART_EXP_PRIO_YOUNG      // timing experiments (tools/ablate_bf16.py): static priority for one half of the waves
    if (wv >= 4) __builtin_amdgcn_s_setprio(1);
#endif
#ifdef NERFART_EXP_PRIO_OLD
    if (wv < 4) __builtin_amdgcn_s_setprio(1);
#endif
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        s.wrap = (tile + gridDim.x) < ntiles;
        const unsigned m = tile * 128u + wv * 16 + j;
        const Pt pt = fetch_point(src, m, false);
        float sdf = surface_chain<0>(pt.x, pt.y, pt.z, g, -1, s, aux, nullptr, true) + aux[SURF_AUX_B8];
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

#ifndef NERFART_K2_ONLY        // mlp_chain_f16x1.hip instantiates K2 alone
// =======================================================================================
// K3a (split bf16): sdf + nabla + h7, forward mode; 32 points per workgroup tile (4 per wave, quads:
// column 4i = value, 4i+1..3 = d/dx, d/dy, d/dz).
// =======================================================================================
__global__ void __launch_bounds__(WG_THREADS, 2)
k_sdf_nabla_bf16(const float* __restrict__ blob, PointSrc src, float R_bg, float* __restrict__ sdf_out,
                 float* __restrict__ nabla_out, float* __restrict__ h7_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int* hdr = reinterpret_cast<const int*>(blob);
    float* aux = smem + 2 * CHUNK_FLOATS;
    const int lane = lane_id(), g = lane >> 4, j = lane & 15, wv = wave_id();
    const int cq = j & 3;
    load_aux(aux, blob, hdr, SURF_AUX_FLOATS);
    const unsigned ntiles = (src.M + 31u) / 32u;
    if (blockIdx.x >= ntiles) return;
    Stream s = make_stream(blob, aux, smem, hdr[2]);
    s.wrap = (blockIdx.x + gridDim.x) < ntiles;
    stream_start(s);
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        s.wrap = (tile + gridDim.x) < ntiles;
        const unsigned m = tile * 32u + wv * 4 + (j >> 2);
        const Pt pt = fetch_point(src, m, false);
        float* h7_lane = (h7_out != nullptr && m < src.M) ? h7_out + (size_t)m * 256 : nullptr;
        const float v = surface_chain<1>(pt.x, pt.y, pt.z, g, cq - 1, s, aux, h7_lane, cq == 0);
        if (m < src.M && g == 0) {
            if (cq == 0) {
                float sdf = v + aux[SURF_AUX_B8];
                if (R_bg > 0.f) {
                    const float d_bg = R_bg - sqrtf(pt.x * pt.x + pt.y * pt.y + pt.z * pt.z);
                    sdf = (d_bg < sdf) ? d_bg : sdf;
                }
                sdf_out[m] = sdf;
            } else {
                nabla_out[(size_t)m * 3 + (cq - 1)] = v;
            }
        }
    }
}

// =======================================================================================
// K3b (split bf16): radiance net.  VE extra units: 1 (VolSDF, 9 extras) or 2 (NeuS, 33 extras).
// =======================================================================================
template <int VE>
__device__ __forceinline__ void radiance_extras(const Pt& pt, float nx, float ny, float nz, int g, Unit (&X)[VE]) {
    constexpr int NE = (VE == 1) ? 9 : 33;
    float ex[VE * 32];
#pragma unroll
    for (int k = 0; k < VE * 32; ++k) ex[k] = 0.f;
    ex[0] = pt.x; ex[1] = pt.y; ex[2] = pt.z;
    const float v[3] = {pt.vx, pt.vy, pt.vz};
#pragma unroll
    for (int c = 0; c < 3; ++c) ex[3 + c] = v[c];
    if (VE == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float sn, co;
                sincosf(v[c] * (float)(1 << k), &sn, &co);
                ex[6 + 6 * k + c] = sn;
                ex[6 + 6 * k + 3 + c] = co;
            }
    }
    ex[NE - 3] = nx; ex[NE - 2] = ny; ex[NE - 1] = nz;
    // slot (q, g, e) <-> extra index 32q + 8g + e
#pragma unroll
    for (int q = 0; q < VE; ++q) {
        u32x4 hi, lo;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const int o = 32 * q + 2 * pr;
            const float a = (g == 0) ? ex[o] : ((g == 1) ? ex[o + 8] : ((g == 2) ? ex[o + 16] : ex[o + 24]));
            const float b = (g == 0) ? ex[o + 1] : ((g == 1) ? ex[o + 9] : ((g == 2) ? ex[o + 17] : ex[o + 25]));
            unsigned sh, sl;
            split2(a, b, sh, sl);
            hi[pr] = sh; lo[pr] = sl;
        }
        X[q].h = hi;
        X[q].l = lo;
    }
}

template <int VE, bool DUMP>
__global__ void __launch_bounds__(WG_THREADS, 2)
k_radiance_bf16(const float* __restrict__ blob, PointSrc src, const float* __restrict__ nabla_in,
                const float* __restrict__ h7_in, float* __restrict__ rgb_out, char* __restrict__ dump) {
    constexpr int M = DUMP ? 8 : 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int* hdr = reinterpret_cast<const int*>(blob);
    float* aux = smem + 2 * CHUNK_FLOATS;
    const int lane = lane_id(), g = lane >> 4, j = lane & 15, wv = wave_id();
    load_aux(aux, blob, hdr, RAD_AUX_FLOATS);
    const unsigned ntiles = (src.M + 127u) / 128u;
    if (blockIdx.x >= ntiles) return;
    Stream s = make_stream(blob, aux, smem, hdr[2]);
    s.wrap = (blockIdx.x + gridDim.x) < ntiles;
    stream_start(s);
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        s.wrap = (tile + gridDim.x) < ntiles;
        const unsigned m = tile * 128u + wv * 16 + j;
        const bool valid = m < src.M;
        const Pt pt = fetch_point(src, m, true);
        float nx = 0.f, ny = 0.f, nz = 0.f;
        if (valid) { nx = nabla_in[(size_t)m * 3 + 0]; ny = nabla_in[(size_t)m * 3 + 1]; nz = nabla_in[(size_t)m * 3 + 2]; }
        Acc A, B;
        Unit x0, x0n, none[1];
        GradCtx gc;
        gc.ws = gc.ws_out = gc.pend_ptr = dump;                       // [5 slots][ntiles * 128 points][256] bf16, point-major
        gc.slot_stride = (size_t)ntiles * 128 * 512;
        gc.unit_stride = 64;
        gc.voff = gc.voff_out = (tile * 128u + wv * 16 + j) * 512u + g * 16;
        {
            // h7 -> units: unit u slot e < 4: feature 32u + 4g + e; e >= 4: 32u + 16 + 4g + (e - 4)
            Unit hu[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                f32x4 lo4 = {0.f, 0.f, 0.f, 0.f}, hi4 = {0.f, 0.f, 0.f, 0.f};
                if (valid) {
                    const float* s0 = h7_in + (size_t)m * 256 + 32 * u + 4 * g;
                    lo4 = *reinterpret_cast<const f32x4*>(s0);
                    hi4 = *reinterpret_cast<const f32x4*>(s0 + 16);
                }
                unsigned sh[4], sl[4];
                split2(lo4[0], lo4[1], sh[0], sl[0]);
                split2(lo4[2], lo4[3], sh[1], sl[1]);
                split2(hi4[0], hi4[1], sh[2], sl[2]);
                split2(hi4[2], hi4[3], sh[3], sl[3]);
                hu[u].h = u32x4{sh[0], sh[1], sh[2], sh[3]};
                hu[u].l = u32x4{sl[0], sl[1], sl[2], sl[3]};
            }
            none[0] = hu[0];
            x0 = hu[0];
            // geometry feature = W8[1:] h7 + b8[1:] (no activation)
            const EpiCtx ec{-INFINITY, -INFINITY, true};
            gc.layer = 0;
            layer<Cfg<M, M, 0, 8, true, false, false>>(B, A, x0, hu, x0n, s, aux, ec, gc);
        }
        x0 = x0n;
        {
            // [feat | x, v, n] -> 256, ReLU
            Unit ex[VE];
            radiance_extras<VE>(pt, nx, ny, nz, g, ex);
            const EpiCtx ec{-INFINITY, 0.f, true};
            gc.layer = 1;
            layer<Cfg<M, M, 8, VE, true, false, DUMP>>(A, B, x0, ex, x0n, s, aux + 256, ec, gc);
        }
        A = B;
        x0 = x0n;
        const EpiCtx ec{0.f, 0.f, true};
#pragma nounroll
        for (int L = 2; L < 4; ++L) {
            gc.layer = L;
            layer<Cfg<M, M, 8, 0, true, false, DUMP>>(A, B, x0, none, x0n, s, aux + L * 256, ec, gc);
            A = B;
            x0 = x0n;
        }
        gc.layer = 4;
        layer<Cfg<M, M, 8, 0, false, false, DUMP>>(A, B, x0, none, x0n, s, aux + 4 * 256, ec, gc);
        float dot[3] = {0.f, 0.f, 0.f};
        last_epilogue<2, 3>(B, aux + RAD_AUX_ROWS, dot, nullptr, ec, DUMP ? gc.ws_out + uoff(gc, 4 * 8) + gc.voff_out : nullptr, gc.unit_stride);
        float c[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) c[n] = sigmoidf_(sum_over_groups(dot[n]) + aux[RAD_AUX_BF + n]);
        if (valid && g < 3) rgb_out[(size_t)m * 3 + g] = (g == 0) ? c[0] : ((g == 1) ? c[1] : c[2]);
    }
}

#endif  // NERFART_K2_ONLY

}  // namespace b16
}  // namespace nerfart

using namespace nerfart;

namespace nerfart {
// Entry points used by mlp_chain.hip's dispatchers when precision = 1 (split bf16).
// the 8-wave K2 (the default; NERFART_K2=w32 selects the one-wave-per-SIMD kernel of mlp_k2_w32.hip)
int sdf_bf16_v1(const float* blob, const PointSrc& s, float R_bg, float* out, int out_stride, hipStream_t st) {
    return b16::launch_chain(0, (long long)s.M, b16::k_sdf_only_bf16, (s.M + 127u) / 128u, st, blob, s, R_bg, out, out_stride);
}
#ifndef NERFART_K2_ONLY
int sdf_nabla_bf16(const float* blob, const PointSrc& s, float R_bg, float* sdf, float* nabla, float* h7, hipStream_t st) {
    return b16::launch_chain(1, (long long)s.M, b16::k_sdf_nabla_bf16, (s.M + 31u) / 32u, st, blob, s, R_bg, sdf, nabla, h7);
}
int radiance_bf16(const float* blob, int view_tiles, const PointSrc& s, const float* nabla, const float* h7, float* rgb, hipStream_t st) {
    const unsigned nt = (s.M + 127u) / 128u;
    if (view_tiles == 1) return b16::launch_chain(2, (long long)s.M, b16::k_radiance_bf16<1, false>, nt, st, blob, s, nabla, h7, rgb, (char*)nullptr);
    if (view_tiles == 3) return b16::launch_chain(2, (long long)s.M, b16::k_radiance_bf16<2, false>, nt, st, blob, s, nabla, h7, rgb, (char*)nullptr);
    set_last_error("radiance_fwd: view_tiles must be 1 (raw view dirs) or 3 (multires_view = 4)");
    return 2;
}
size_t radiance_dump_bytes(long long M) { return (size_t)((M + 127) / 128) * b16::RAD_DUMP_PER_TILE; }
int radiance_fwd_dump_bf16(const float* blob, int view_tiles, const PointSrc& s, const float* nabla, const float* h7, float* rgb, void* dump, hipStream_t st) {
    const unsigned nt = (s.M + 127u) / 128u;
    if (s.M > (1u << 21)) { set_last_error("radiance_fwd_dump: at most 2^21 points per call"); return 2; }
    if (view_tiles == 1) return b16::launch_chain(2, (long long)s.M, b16::k_radiance_bf16<1, true>, nt, st, blob, s, nabla, h7, rgb, (char*)dump);
    if (view_tiles == 3) return b16::launch_chain(2, (long long)s.M, b16::k_radiance_bf16<2, true>, nt, st, blob, s, nabla, h7, rgb, (char*)dump);
    set_last_error("radiance_fwd_dump: view_tiles must be 1 or 3");
    return 2;
}
#endif  // NERFART_K2_ONLY

}  // namespace nerfart
