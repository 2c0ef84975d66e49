// mlp_backward_bf16.hip - the per-sample backward kernels of the fine-tune step (row a19): second-order backward of the
// SDF net (k_sdf_fwd2_bf16 / k_sdf_bwd2_bf16) and the radiance net's backward (k_radiance_bwd_bf16).
#include "mlp_bf16_core.h"

namespace nerfart {
namespace b16 {

// =======================================================================================
// Second-order backward of the SDF net (row a19): parameter gradients of
//     Phi = sbar * sdf + hbar7 . a_7 + nbar . grad_x sdf            (sbar, hbar7, nbar: cotangents, per point)
// nbar . grad_x sdf is the directional derivative of sdf along nbar, i.e. w8 . adot_7 with adot the forward-mode
// tangent; reverse mode over the (value, tangent) recursion gives, per layer,
//     t_{l-1} = W_l^T (t_l * d_l)                                    (t_7 = w8: the reverse sweep of k_sdf_grad_bf16)
//     abar_{l-1} = W_l^T zbar_l,  zbar_l = abar_l * d_l + 100 t_l adot_l (1 - d_l)      (softplus'' = 100 d (1 - d))
//     dW_l = sum_points zbar_l (x) a_{l-1} + (t_l d_l) (x) adot_{l-1},   db_l = sum zbar_l
// Two kernels with COLUMN PAIRS (8 points per wave), both dumping bf16 GEMM operands as POINT-MAJOR matrices the library
// GEMMs read in place: [slot][2 Mp rows][256 features in unit order], Mp = 64 * tiles; rows 0..Mp-1 of a slot come from one
// column of each pair and rows Mp.. from the other (GradCtx in mlp_bf16_core.h):
//   k_sdf_fwd2_bf16: columns (value, tangent along nbar): slots 0..7 = [a_l; adot_l], slots 8..15 = softplus'(z_l) unorm16
//   k_sdf_bwd2_bf16: columns (t, abar): consumes those, slots 0..7 = 65535 * [zbar_l; t_l d_l]
// The weight-gradient GEMMs and the weight_norm chain rule are host side (autodiff.SurfaceBackward).
// =======================================================================================
constexpr int F2_DUMP_PER_TILE = 16 * 8 * 8 * 1024;
constexpr int R2_DUMP_PER_TILE = 8 * 8 * 8 * 1024;

// encoding units for the (value, tangent) pairs: value lanes as encode_units, tangent lanes d enc / d x . dir
__device__ __forceinline__ void encode_units_pair(float x, float y, float z, float dx, float dy, float dz, int g, bool is_val, Unit (&X)[2]) {
    const float cg = (g == 0) ? x : ((g == 1) ? y : z);
    const float wg = (g == 0) ? dx : ((g == 1) ? dy : dz);
    const bool live = g < 3;
    float m[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) m[k] = 0.f;
    m[0] = is_val ? cg : wg;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float f = (float)(1 << k);
        float sn, cs;
        sincosf(cg * f, &sn, &cs);
        m[1 + 2 * k] = is_val ? sn : cs * f * wg;
        m[2 + 2 * k] = is_val ? cs : -(sn * f) * wg;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        u32x4 hi, lo;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const float a = live ? m[8 * q + 2 * pr] : 0.f, b = live ? m[8 * q + 2 * pr + 1] : 0.f;
            unsigned sh, sl;
            split2(a, b, sh, sl);
            hi[pr] = sh; lo[pr] = sl;
        }
        X[q].h = hi;
        X[q].l = lo;
    }
}

__global__ void __launch_bounds__(WG_THREADS, 2)
k_sdf_fwd2_bf16(const float* __restrict__ blob, unsigned M, const float* __restrict__ pts, const float* __restrict__ dirv,
                char* __restrict__ dump) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int* hdr = reinterpret_cast<const int*>(blob);
    float* aux = smem + 2 * CHUNK_FLOATS;
    const int lane = lane_id(), g = lane >> 4, j = lane & 15, wv = wave_id();
    const bool is_val = (j & 1) == 0;
    load_aux(aux, blob, hdr, SURF_AUX_FLOATS);
    const unsigned ntiles = (M + 63u) / 64u;
    if (blockIdx.x >= ntiles) return;
    Stream s = make_stream(blob, aux, smem, hdr[2]);
    s.wrap = (blockIdx.x + gridDim.x) < ntiles;
    stream_start(s);
    const EpiCtx ec{0.f, 0.f, is_val};
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        s.wrap = (tile + gridDim.x) < ntiles;
        const unsigned m = tile * 64u + wv * 8 + (j >> 1);
        float px = 0.f, py = 0.f, pz = 0.f, vx = 0.f, vy = 0.f, vz = 0.f;
        if (m < M) {
            px = pts[3 * (size_t)m]; py = pts[3 * (size_t)m + 1]; pz = pts[3 * (size_t)m + 2];
            vx = dirv[3 * (size_t)m]; vy = dirv[3 * (size_t)m + 1]; vz = dirv[3 * (size_t)m + 2];
        }
        GradCtx gc;
        gc.ws = gc.ws_out = gc.pend_ptr = dump;        // [16 slots][2 Mp rows: value columns, then tangent columns][256] bf16
        gc.slot_stride = (size_t)ntiles * 128 * 512;
        gc.unit_stride = 64;
        gc.voff = gc.voff_out = (((j & 1) ? ntiles * 64u : 0u) + tile * 64u + wv * 8 + (j >> 1)) * 512u + g * 16;
        gc.voff_st2 = is_val ? gc.voff_out : 0xffffff00u;   // softplus' slots (8..15): value rows only, the reverse sweep reads those
        gc.st2_bytes = ntiles * 64u * 512u - 448u;          // (a unit base is up to 448 B into row 0: the range ends with the value rows)
        // unit 7 of layer 3 (features 224..255 of a 217-wide layer) is never built: the reverse sweep still reads it (against
        // zero weights) - it must not hold a NaN
        *reinterpret_cast<u32x4*>(gc.ws_out + uoff(gc, 3 * 8 + 7) + gc.voff_out) = u32x4{0u, 0u, 0u, 0u};
        st2_store(gc, gc.ws_out + uoff(gc, 64 + 3 * 8 + 7), u32x4{0u, 0u, 0u, 0u});
        Acc A, B;
        Unit x0, x0n, enc[2], none[1];
        encode_units_pair(px, py, pz, vx, vy, vz, g, is_val, enc);
        none[0] = enc[0];
        x0 = enc[0];
        gc.layer = 0;
        layer<Cfg<10, 10, 0, 2, true, false, false>>(B, A, x0, enc, x0n, s, aux, ec, gc);
        x0 = x0n;
#pragma nounroll
        for (int L = 1; L < 7; ++L) {
            gc.layer = L;
            if (L == 4) {
                encode_units_pair(px, py, pz, vx, vy, vz, g, is_val, enc);
                layer<Cfg<10, 10, 7, 2, true, false, true>>(A, B, x0, enc, x0n, s, aux + L * 256, ec, gc);
            } else {
                layer<Cfg<10, 10, 8, 0, true, false, true>>(A, B, x0, none, x0n, s, aux + L * 256, ec, gc);
            }
            A = B;
            x0 = x0n;
        }
        gc.layer = 7;
        layer<Cfg<10, 10, 8, 0, false, false, true>>(A, B, x0, none, x0n, s, aux + 7 * 256, ec, gc);
        // layer 7's own activations / tangents / softplus' (no later layer hosts them)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            u32x4 hi, dd;
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                const int r0 = 2 * (pr & 1);
                const f32x4 t = (pr >> 1) ? B.t[2 * u + 1] : B.t[2 * u];
                Work w = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                unsigned h = 0, l = 0, dout = 0;
                epi_phase<10, 0>(t[r0], t[r0 + 1], w, h, l, 0.f, is_val, 0u, dout);
                epi_phase<10, 1>(t[r0], t[r0 + 1], w, h, l, 0.f, is_val, 0u, dout);
                epi_phase<10, 2>(t[r0], t[r0 + 1], w, h, l, 0.f, is_val, 0u, dout);
                hi[pr] = h; dd[pr] = dout;
            }
            *reinterpret_cast<u32x4*>(gc.ws_out + uoff(gc, 7 * 8 + u) + gc.voff_out) = hi;
            st2_store(gc, gc.ws_out + uoff(gc, 64 + 7 * 8 + u), dd);
        }
    }
}

// unit u of the reverse sweep's input, outside the hosted pipeline (first unit of layer 7, all units of layer 0)
__device__ __forceinline__ Unit pair_unit(const Acc& P, int u, const u32x4 dd, const u32x4 aa, bool is_val, u32x4& hi_out) {
    Unit X;
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
        const int r0 = 2 * (pr & 1);
        const f32x4 t = (pr >> 1) ? P.t[2 * u + 1] : P.t[2 * u];
        Work w = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        unsigned h = 0, l = 0, dout = 0;
        epi_phase<11, 0>(t[r0], t[r0 + 1], w, h, l, 0.f, is_val, dd[pr], dout, aa[pr]);
        epi_phase<11, 1>(t[r0], t[r0 + 1], w, h, l, 0.f, is_val, dd[pr], dout, aa[pr]);
        epi_phase<11, 2>(t[r0], t[r0 + 1], w, h, l, 0.f, is_val, dd[pr], dout, aa[pr]);
        X.h[pr] = h; X.l[pr] = l;
    }
    hi_out = X.h;
    return X;
}

__global__ void __launch_bounds__(WG_THREADS, 2)
k_sdf_bwd2_bf16(const float* __restrict__ blob, unsigned M, const float* __restrict__ gbar_h7, const float* __restrict__ gbar_sdf,
                char* __restrict__ f2_dump, char* __restrict__ r2_dump) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int* hdr = reinterpret_cast<const int*>(blob);
    float* aux = smem + 2 * CHUNK_FLOATS;
    const int lane = lane_id(), g = lane >> 4, j = lane & 15, wv = wave_id();
    const bool is_val = (j & 1) == 0;                       // even lanes: t = d sdf / d a; odd lanes: abar
    {   // aux + the chunk table of this program (header words 128..)
        const float* src = blob + hdr[4];
        for (int i = threadIdx.x; i < SURF_AUX_FLOATS; i += WG_THREADS) aux[i] = src[i];
        int* tab = reinterpret_cast<int*>(aux + AUX_FLOATS_MAX);
        if (threadIdx.x < TAB_INTS) tab[threadIdx.x] = hdr[128 + threadIdx.x];
        __syncthreads();
    }
    const unsigned ntiles = (M + 63u) / 64u;
    if (blockIdx.x >= ntiles) return;
    Stream s = make_stream(blob, aux, smem, hdr[7]);
    s.wrap = (blockIdx.x + gridDim.x) < ntiles;
    stream_start(s);
    const EpiCtx ec{0.f, 0.f, is_val};
    const float* row = aux + SURF_AUX_ROW;
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        s.wrap = (tile + gridDim.x) < ntiles;
        const unsigned m = tile * 64u + wv * 8 + (j >> 1);
        const bool valid = m < M;
        GradCtx gc;
        gc.ws = f2_dump;
        gc.ws_out = gc.pend_ptr = r2_dump;             // [8 slots][2 Mp rows: abar columns (zbar), then t columns (t d)][256] bf16
        gc.slot_stride = (size_t)ntiles * 128 * 512;
        gc.unit_stride = 64;
        {
            const unsigned pt = tile * 64u + wv * 8 + (j >> 1);
            // both lanes of a pair read the point's softplus' from its value row and its tangent from its tangent row (the
            // activations in the value rows and a second copy of softplus' are not needed here): 8 KiB per point instead of 16
            gc.voff = pt * 512u + g * 16;
            gc.voff2 = (ntiles * 64u + pt) * 512u + g * 16;
            gc.voff_out = (((j & 1) ? 0u : ntiles * 64u) + pt) * 512u + g * 16;
        }
        const float sb = valid ? gbar_sdf[m] : 0.f;
        Acc A, B;
#pragma unroll
        for (int T = 0; T < 16; ++T) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(row + 16 * T + 4 * g);
            f32x4 hb = {0.f, 0.f, 0.f, 0.f};
            if (valid && !is_val) hb = *reinterpret_cast<const f32x4*>(gbar_h7 + (size_t)m * 256 + 16 * T + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) A.t[T][r] = is_val ? w0[r] : fmaf(sb, w0[r], hb[r]);
        }
        Unit x0, x0n, none[1];
        {
            const u32x4 dd = *reinterpret_cast<const u32x4*>(gc.ws + uoff(gc, 64 + 7 * 8) + gc.voff);
            const u32x4 aa = *reinterpret_cast<const u32x4*>(gc.ws + uoff(gc, 7 * 8) + gc.voff2);
            u32x4 hi;
            x0 = pair_unit(A, 0, dd, aa, is_val, hi);
            *reinterpret_cast<u32x4*>(gc.ws_out + uoff(gc, 7 * 8) + gc.voff_out) = hi;
        }
        none[0] = x0;
        d_load2(gc, 7 * 8 + 1, 0);
        gc.layer = 7;
        layer<Cfg<11, 11, 8, 0, true, true, false, true>>(A, B, x0, none, x0n, s, aux, ec, gc);
        A = B;
        x0 = x0n;
#pragma nounroll
        for (int L = 6; L > 0; --L) {
            gc.layer = L;
            layer<Cfg<11, 11, 8, 0, true, true, true, true>>(A, B, x0, none, x0n, s, aux, ec, gc);
            A = B;
            x0 = x0n;
        }
        // A = (t_0 | abar_0): the deltas of layer 0 (unit 0 was built by the last step and is still pending)
        *reinterpret_cast<u32x4*>(gc.pend_ptr + gc.voff_out) = gc.dpend;
#pragma unroll
        for (int u = 1; u < 8; ++u) {
            const u32x4 dd = *reinterpret_cast<const u32x4*>(gc.ws + uoff(gc, 64 + u) + gc.voff);
            const u32x4 aa = *reinterpret_cast<const u32x4*>(gc.ws + uoff(gc, u) + gc.voff2);
            u32x4 hi;
            (void)pair_unit(A, u, dd, aa, is_val, hi);
            *reinterpret_cast<u32x4*>(gc.ws_out + uoff(gc, u) + gc.voff_out) = hi;
        }
    }
}

// =======================================================================================
// Radiance net, backward (row a19): cotangents of the layer-7 activation h7 and of the normal input, and the
// per-layer deltas for the weight-gradient GEMMs, from d loss / d rgb.  Consumes the activations the forward
// kernel dumped (k_radiance_bf16<VE, true>: f = geometry feature, r0..r3 = relu outputs, bf16 hi parts in unit
// order) - only their signs here - and streams the transposed-weight chunks that follow the forward program in the
// blob (packing.radiance_plan_bf16): R3^T, R2^T, R1^T, the normal rows of R0^T, the feature rows of R0^T, W8[1:]^T.
// Dumps (bf16, point-major [slot][128 * tiles points][256 in unit order] like the forward dump): delta0 .. delta3 (in the
// slots of the activations f, r0, r1, r2 they are multiplied with), g_f.
// =======================================================================================
template <int IT>
struct NrmItems {      // the 3 normal rows of R0^T: 8 k-steps x 1 output tile from one 16 KiB chunk
    static __device__ __forceinline__ void run(f32x4& E, const Unit (&X)[8], Ring3& r, unsigned addr, const Stream& s) {
        constexpr int N = 8;
        if constexpr (IT < N) {
            constexpr int S = IT % 3, S2 = (IT + 2) % 3;
            constexpr int PENDING = (IT + 2 < N) ? 4 : ((IT + 1 < N) ? 2 : 0);
            if constexpr (IT + 2 < N) {
                if constexpr (S2 == 0) lds_read_pair<(IT + 2) * 2048>(r.h0, r.l0, addr);
                else if constexpr (S2 == 1) lds_read_pair<(IT + 2) * 2048>(r.h1, r.l1, addr);
                else lds_read_pair<(IT + 2) * 2048>(r.h2, r.l2, addr);
            }
            if constexpr (S == 0) { lds_wait_pair<PENDING>(r.h0, r.l0); E = mfma3(r.h0, r.l0, X[IT].h, X[IT].l, E); }
            else if constexpr (S == 1) { lds_wait_pair<PENDING>(r.h1, r.l1); E = mfma3(r.h1, r.l1, X[IT].h, X[IT].l, E); }
            else { lds_wait_pair<PENDING>(r.h2, r.l2); E = mfma3(r.h2, r.l2, X[IT].h, X[IT].l, E); }
            stream_piece<IT>(s);
            __builtin_amdgcn_sched_barrier(0);
            NrmItems<IT + 1>::run(E, X, r, addr, s);
        }
    }
};

// unit u of (P * [mask != 0]) as a B operand
__device__ __forceinline__ Unit masked_unit(const Acc& P, int u, const u32x4 mk) {
    Unit X;
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
        const int r0 = 2 * (pr & 1);
        const f32x4 t = (pr >> 1) ? P.t[2 * u + 1] : P.t[2 * u];
        const float y0 = (mk[pr] & 0xffffu) ? t[r0] : 0.f;
        const float y1 = (mk[pr] >> 16) ? t[r0 + 1] : 0.f;
        unsigned hi, lo;
        split2(y0, y1, hi, lo);
        X.h[pr] = hi; X.l[pr] = lo;
    }
    return X;
}

__global__ void __launch_bounds__(WG_THREADS, 2)
k_radiance_bwd_bf16(const float* __restrict__ blob, unsigned M, const float* __restrict__ rgb, const float* __restrict__ g_rgb,
                    char* __restrict__ fwd_dump, char* __restrict__ bwd_dump, float* __restrict__ g_h7_out,
                    float* __restrict__ g_n_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int* hdr = reinterpret_cast<const int*>(blob);
    float* aux = smem + 2 * CHUNK_FLOATS;
    const int lane = lane_id(), g = lane >> 4, j = lane & 15, wv = wave_id();
    load_aux(aux, blob, hdr, RAD_AUX_FLOATS);
    const unsigned ntiles = (M + 127u) / 128u;
    if (blockIdx.x >= ntiles) return;
    Stream s = make_stream(blob, aux, smem, hdr[6] - hdr[2]);          // the reverse chunks only
    s.tab += hdr[2];
    s.wrap = (blockIdx.x + gridDim.x) < ntiles;
    stream_start(s);
    const EpiCtx ec{0.f, 0.f, true};
    const float* rows = aux + RAD_AUX_ROWS;
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        s.wrap = (tile + gridDim.x) < ntiles;
        const unsigned m = tile * 128u + wv * 16 + j;
        const bool valid = m < M;
        GradCtx gc;
        gc.ws = fwd_dump;
        gc.ws_out = gc.pend_ptr = bwd_dump;
        gc.slot_stride = (size_t)ntiles * 128 * 512;
        gc.unit_stride = 64;
        gc.voff = gc.voff_out = (tile * 128u + wv * 16 + j) * 512u + g * 16;
        // delta4 = g_rgb * rgb (1 - rgb); g_r3 = R4^T delta4 (3 rows: plain FMAs)
        float d4[3] = {0.f, 0.f, 0.f};
        if (valid) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float o = rgb[(size_t)m * 3 + c];
                d4[c] = g_rgb[(size_t)m * 3 + c] * o * (1.f - o);
            }
        }
        Acc A, B;
#pragma unroll
        for (int T = 0; T < 16; ++T) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(rows + 16 * T + 4 * g);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(rows + 256 + 16 * T + 4 * g);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(rows + 512 + 16 * T + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) A.t[T][r] = fmaf(d4[2], w2[r], fmaf(d4[1], w1[r], d4[0] * w0[r]));
        }
        Unit x0, x0n, none[1];
        {   // unit 0 of delta3 (mask = r3, slot 4); it is also the first dumped unit (delta slot 3)
            const u32x4 mk = *reinterpret_cast<const u32x4*>(gc.ws + uoff(gc, 4 * 8) + gc.voff);
            x0 = masked_unit(A, 0, mk);
            *reinterpret_cast<u32x4*>(gc.ws_out + uoff(gc, 3 * 8) + gc.voff_out) = x0.h;
        }
        none[0] = x0;
        d_load(gc, 4 * 8 + 1, 0);
        gc.layer = 4;
        layer<Cfg<7, 7, 8, 0, true, true, false, true>>(A, B, x0, none, x0n, s, aux, ec, gc);       // R3^T
        A = B;
        x0 = x0n;
#pragma nounroll
        for (int L = 3; L > 1; --L) {
            gc.layer = L;
            layer<Cfg<7, 7, 8, 0, true, true, true, true>>(A, B, x0, none, x0n, s, aux, ec, gc);     // R2^T, R1^T
            A = B;
            x0 = x0n;
        }
        // A = g_r0.  Normal rows of R0^T (d loss / d n) need all 8 units of delta0 at once.
        {
            Unit X[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const u32x4 mk = *reinterpret_cast<const u32x4*>(gc.ws + uoff(gc, 1 * 8 + u) + gc.voff);
                X[u] = masked_unit(A, u, mk);
            }
            const float* wp = stream_acquire(s) + lane * 4;
            const unsigned addr = (unsigned)(size_t)wp;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            Ring3 r;
            f32x4 E = {0.f, 0.f, 0.f, 0.f};
            lds_read_pair<0>(r.h0, r.l0, addr);
            lds_read_pair<2048>(r.h1, r.l1, addr);
            NrmItems<0>::run(E, X, r, addr, s);
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
            if (valid && g == 0) { g_n_out[(size_t)m * 3] = E[0]; g_n_out[(size_t)m * 3 + 1] = E[1]; g_n_out[(size_t)m * 3 + 2] = E[2]; }
        }
        gc.layer = 1;
        layer<Cfg<7, 9, 8, 0, true, true, true, false>>(A, B, x0, none, x0n, s, aux, ec, gc);        // feature rows of R0^T
        A = B;
        x0 = x0n;
        gc.layer = 0;
        layer<Cfg<9, 9, 8, 0, false, true, true, false>>(A, B, x0, none, x0n, s, aux, ec, gc);       // W8[1:]^T
        if (valid) {
            float* o = g_h7_out + (size_t)m * 256;
#pragma unroll
            for (int T = 0; T < 16; ++T) *reinterpret_cast<f32x4*>(o + 16 * T + 4 * g) = B.t[T];
        }
    }
}


}  // namespace b16
}  // namespace nerfart

using namespace nerfart;

namespace nerfart {
size_t sdf_fwd2_dump_bytes(long long M) { return (size_t)((M + 63) / 64) * b16::F2_DUMP_PER_TILE; }
size_t sdf_bwd2_dump_bytes(long long M) { return (size_t)((M + 63) / 64) * b16::R2_DUMP_PER_TILE; }
// the per-lane byte offsets into the point-major dumps are 32 bit: rows * 512 B < 4 GiB
static constexpr long long DUMP_MAX_POINTS = 1ll << 21;
int sdf_fwd2_bf16(const float* blob, long long M, const float* pts, const float* dirv, void* dump, hipStream_t st) {
    if (M > DUMP_MAX_POINTS) { set_last_error("second-order / backward kernels: at most 2^21 points per call"); return 2; }
    return b16::launch_chain(-1, M, b16::k_sdf_fwd2_bf16, (unsigned)((M + 63) / 64), st, blob, (unsigned)M, pts, dirv, (char*)dump);
}
int sdf_bwd2_bf16(const float* blob, long long M, const float* gbar_h7, const float* gbar_sdf, void* f2_dump, void* r2_dump, hipStream_t st) {
    if (M > DUMP_MAX_POINTS) { set_last_error("second-order / backward kernels: at most 2^21 points per call"); return 2; }
    return b16::launch_chain(-1, M, b16::k_sdf_bwd2_bf16, (unsigned)((M + 63) / 64), st, blob, (unsigned)M, gbar_h7, gbar_sdf, (char*)f2_dump,
                             (char*)r2_dump);
}
int radiance_bwd_bf16(const float* blob, long long M, const float* rgb, const float* g_rgb, void* fwd_dump, void* bwd_dump, float* g_h7,
                      float* g_n, hipStream_t st) {
    if (M > DUMP_MAX_POINTS) { set_last_error("second-order / backward kernels: at most 2^21 points per call"); return 2; }
    return b16::launch_chain(-1, M, b16::k_radiance_bwd_bf16, (unsigned)((M + 127) / 128), st, blob, (unsigned)M, rgb, g_rgb, (char*)fwd_dump,
                             (char*)bwd_dump, g_h7, g_n);
}

}  // namespace nerfart
