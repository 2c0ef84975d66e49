// mlp_grad_bf16.hip - reverse-mode grad(SDF): what nerfart_sdf_nabla_fwd runs at precision 1 (split bf16).
#include "mlp_bf16_core.h"

namespace nerfart {
namespace b16 {

// =======================================================================================
// K3a, reverse mode (split bf16): sdf + nabla + h7 with ONE column per point.  Forward sweep = K2 plus
// softplus'(z_l) of layers 0..6 to the scratch; then d sdf / d a_{l-1} = W_l^T (d sdf / d a_l * softplus'(z_l))
// for l = 7..0 through the transposed-weight chunks that follow the forward program in the blob
// (packing.surface_plan_bf16): 1.97 M algorithmic flop per point instead of the 4.2 M of the forward-mode
// quads, and a cheap multiply epilogue.  Rows 217..255 of the layer-4 and layer-0 steps are d sdf / d enc
// (3 accumulator tiles kept from layer 4 to the end) and meet the encoding's Jacobian in registers.
// =======================================================================================
constexpr int GRAD_WS_PER_WG = 7 * 8 * 8 * 1024;           // 7 layers x 8 units x 8 waves x 1 KiB

template <int IT>
struct TailItems {     // layer-0 step: 8 k-steps x 3 output tiles from one 48 KiB chunk
    static __device__ __forceinline__ void run(f32x4 (&E)[3], const Unit (&X)[8], Ring3& r, unsigned addr, const Stream& s) {
        constexpr int N = 24;
        if constexpr (IT < N) {
            constexpr int ks = IT / 3, t = IT % 3;
            constexpr int S = IT % 3, S2 = (IT + 2) % 3;
            constexpr int PENDING = (IT + 2 < N) ? 4 : ((IT + 1 < N) ? 2 : 0);
            if constexpr (IT + 2 < N) {
                if constexpr (S2 == 0) lds_read_pair<(IT + 2) * 2048>(r.h0, r.l0, addr);
                else if constexpr (S2 == 1) lds_read_pair<(IT + 2) * 2048>(r.h1, r.l1, addr);
                else lds_read_pair<(IT + 2) * 2048>(r.h2, r.l2, addr);
            }
            if constexpr (S == 0) { lds_wait_pair<PENDING>(r.h0, r.l0); E[t] = mfma3(r.h0, r.l0, X[ks].h, X[ks].l, E[t]); }
            else if constexpr (S == 1) { lds_wait_pair<PENDING>(r.h1, r.l1); E[t] = mfma3(r.h1, r.l1, X[ks].h, X[ks].l, E[t]); }
            else { lds_wait_pair<PENDING>(r.h2, r.l2); E[t] = mfma3(r.h2, r.l2, X[ks].h, X[ks].l, E[t]); }
            if constexpr (IT < 8) stream_piece<IT>(s);
            __builtin_amdgcn_sched_barrier(0);
            TailItems<IT + 1>::run(E, X, r, addr, s);
        }
    }
};

__global__ void __launch_bounds__(WG_THREADS, 2)
k_sdf_grad_bf16(const float* __restrict__ blob, PointSrc src, float R_bg, float* __restrict__ sdf_out,
                float* __restrict__ nabla_out, float* __restrict__ h7_out, char* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int* hdr = reinterpret_cast<const int*>(blob);
    float* aux = smem + 2 * CHUNK_FLOATS;
    const int lane = lane_id(), g = lane >> 4, j = lane & 15, wv = wave_id();
    load_aux(aux, blob, hdr, SURF_AUX_FLOATS);
    const unsigned ntiles = (src.M + 127u) / 128u;
    if (blockIdx.x >= ntiles) return;
    Stream s = make_stream(blob, aux, smem, hdr[6]);          // forward + backward chunks
    s.wrap = (blockIdx.x + gridDim.x) < ntiles;
    stream_start(s);
    GradCtx gc;
    // the workgroup's 448 KiB of scratch as one buffer: descriptor from kernel arguments and blockIdx only (provably uniform),
    // wave, lane and the unit's offset (slot * 64 KiB + unit * 8 KiB) together in the 32-bit voffset
#ifdef NERFART_NO_BUF_SCRATCH        // A/B only (tools/ab_variant.py): the round-2 addressing, a 64-bit lane pointer per unit
    gc.buf = false;
    gc.ws = gc.ws_out = ws + (size_t)blockIdx.x * GRAD_WS_PER_WG;
#else
    gc.buf = true;
    gc.rsrc = __builtin_amdgcn_make_buffer_rsrc(ws + (size_t)blockIdx.x * GRAD_WS_PER_WG, 0, GRAD_WS_PER_WG, 0x00020000);
    gc.ws = gc.ws_out = nullptr;
#endif
#ifdef NERFART_EXP_SCRATCH_SMALL    // timing experiment: every unit lands on the same L2-resident KiB (results wrong)
    gc.slot_stride = 0;
    gc.unit_stride = 0;
#else
    gc.slot_stride = 8 * 8192;
    gc.unit_stride = 8192;
#endif
    gc.voff = gc.voff_out = wv * 1024 + lane * 16;
    gc.pend_ptr = nullptr;
    gc.pend_soff = 0;
    const EpiCtx ec{0.f, 0.f, true};
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        s.wrap = (tile + gridDim.x) < ntiles;
        const unsigned m = tile * 128u + wv * 16 + j;
        const Pt pt = fetch_point(src, m, false);
        Acc A, B;
        Unit x0, x0n, enc[2], none[1];
        // ---------------- forward sweep ----------------
        encode_units(pt.x, pt.y, pt.z, g, -1, enc);
        none[0] = enc[0];
        x0 = enc[0];
        gc.layer = 0;
        layer<Cfg<5, 5, 0, 2, true, false, false>>(B, A, x0, enc, x0n, s, aux, ec, gc);
        x0 = x0n;
#pragma nounroll
        for (int L = 1; L < 7; ++L) {
            gc.layer = L;
            if (L == 4) {
                encode_units(pt.x, pt.y, pt.z, g, -1, enc);
                layer<Cfg<5, 5, 7, 2, true, false, true>>(A, B, x0, enc, x0n, s, aux + L * 256, ec, gc);
            } else {
                layer<Cfg<5, 5, 8, 0, true, false, true>>(A, B, x0, none, x0n, s, aux + L * 256, ec, gc);
            }
            A = B;
            x0 = x0n;
        }
        gc.layer = 7;
        // layer 7: its last k-step prepares unit 0 of softplus'(z_7) for the first backward step
        layer<Cfg<5, 4, 8, 0, true, false, true>>(A, B, x0, none, x0n, s, aux + 7 * 256, ec, gc);
        x0 = x0n;
        {
            float dot[1] = {0.f};
            float* h7_lane = (h7_out != nullptr && m < src.M) ? h7_out + (size_t)m * 256 : nullptr;
            last_epilogue<0, 1>(B, aux + SURF_AUX_ROW, dot, h7_lane, ec);
            float sdf = sum_over_groups(dot[0]) + aux[SURF_AUX_B8];
            if (R_bg > 0.f) {
                const float d_bg = R_bg - sqrtf(pt.x * pt.x + pt.y * pt.y + pt.z * pt.z);
                sdf = (d_bg < sdf) ? d_bg : sdf;
            }
            if (g == 0 && m < src.M) sdf_out[m] = sdf;
        }
        // ---------------- backward sweep: B = z_7 ----------------
        layer<Cfg<4, 3, 8, 0, true, true, false, true>>(B, A, x0, none, x0n, s, aux, ec, gc);
        x0 = x0n;
        f32x4 E[3];
#pragma nounroll
        for (int L = 6; L > 1; --L) {
            gc.layer = L;
            layer<Cfg<3, 3, 8, 0, true, true, false, true>>(A, B, x0, none, x0n, s, aux, ec, gc);
            if (L == 4) { E[0] = B.t[13]; E[1] = B.t[14]; E[2] = B.t[15]; }
            A = B;
            x0 = x0n;
        }
        gc.layer = 1;
        layer<Cfg<3, 3, 8, 0, false, true, false, false>>(A, B, x0, none, x0n, s, aux, ec, gc);
        // ---------------- layer 0: only the 39 encoding rows, on top of layer 4's ----------------
        {
            u32x4 d0[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#if defined(NERFART_ABLATE_SCRATCH) || defined(NERFART_ABLATE_SCRATCH_LD)
                d0[u] = u32x4{1u, 1u, 1u, 1u};
#else
                d0[u] = unit_load(gc, u);
#endif
            }
            Unit X[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    const int tile = 2 * u + (pr >> 1), r0 = 2 * (pr & 1);
#ifdef NERFART_F16X2      // the fp16 blob does not absorb the unorm16 scale (mlp_bf16_core.h, MODE 3)
                    const float y0 = B.t[tile][r0] * ((float)(d0[u][pr] & 0xffffu) * (1.0f / 65535.0f));
                    const float y1 = B.t[tile][r0 + 1] * ((float)(d0[u][pr] >> 16) * (1.0f / 65535.0f));
#else
                    const float y0 = B.t[tile][r0] * (float)(d0[u][pr] & 0xffffu);
                    const float y1 = B.t[tile][r0 + 1] * (float)(d0[u][pr] >> 16);
#endif
                    unsigned hi, lo;
                    split2(y0, y1, hi, lo);
                    X[u].h[pr] = hi; X[u].l[pr] = lo;
                }
            }
            const float* wp = stream_acquire(s) + lane * 4;
            const unsigned addr = (unsigned)(size_t)wp;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            Ring3 r;
            lds_read_pair<0>(r.h0, r.l0, addr);
            lds_read_pair<2048>(r.h1, r.l1, addr);
            TailItems<0>::run(E, X, r, addr, s);
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        }
        // d sdf / d x = J_enc^T E: lane (g, j), tile t, reg r holds d sdf / d enc[f], f = 16 t + 4 g + r - 9
        float part[3] = {0.f, 0.f, 0.f};
        // the feature bookkeeping below depends on the lane only: hipcc hoisted all of it out of the tile loop (12 x frequency,
        // component and validity masks = ~20 VGPRs and ~60 SGPR pairs held across the whole tile) and spilled it; an opaque copy of
        // the lane group keeps the ~180 integer instructions here, where nothing else is live
        int gj = g;
        asm volatile("" : "+v"(gj));
#pragma unroll
        for (int t = 0; t < 3; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * t + 4 * gj + r - 9;
                const bool ok = (f >= 0) && (f < 39);
                const int q = (f >= 3) ? f - 3 : 0;
                const int k = q / 6, rem = q - 6 * k;
                const int isc = rem / 3;
                const int c = (f < 3) ? f : rem - 3 * isc;
                const float xc = (c == 0) ? pt.x : ((c == 1) ? pt.y : pt.z);
                const float fr = (float)(1 << k);
                float sn, cs;
                sincosf(xc * fr, &sn, &cs);
                float jac = (f < 3) ? 1.f : ((isc == 0) ? cs * fr : -(sn * fr));
                jac = ok ? jac : 0.f;
                const float v = E[t][r] * jac;
                part[0] += (c == 0) ? v : 0.f;
                part[1] += (c == 1) ? v : 0.f;
                part[2] += (c == 2) ? v : 0.f;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) part[c] = sum_over_groups(part[c]);
        if (m < src.M && g < 3) nabla_out[(size_t)m * 3 + g] = (g == 0) ? part[0] : ((g == 1) ? part[1] : part[2]);
    }
}


}  // namespace b16
}  // namespace nerfart

using namespace nerfart;

namespace nerfart {
size_t sdf_grad_ws_bytes() { return (size_t)num_cus() * b16::GRAD_WS_PER_WG; }
int sdf_grad_bf16(const float* blob, const PointSrc& s, float R_bg, float* sdf, float* nabla, float* h7, void* ws, hipStream_t st) {
    return b16::launch_chain(1, (long long)s.M, b16::k_sdf_grad_bf16, (s.M + 127u) / 128u, st, blob, s, R_bg, sdf, nabla, h7, (char*)ws);
}

}  // namespace nerfart
