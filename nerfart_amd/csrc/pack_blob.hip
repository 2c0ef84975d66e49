// pack_blob.hip - the weight-blob packers behind the C ABI (boundary B5: what turns a checkpoint's `weight_g / weight_v / bias` tensors,
// reference models/base.py:226-227 / render.py:266-267, into the blobs every kernel of this library reads).
//
//   nerfart_pack_surface_blob   SDF net (ImplicitSurface, base.py:131-263)                     -> program 1 (fp32) / 3 (split bf16 or fp16)
//   nerfart_pack_radiance_blob  geometry-feature rows of its last layer + RadianceNet (:312-391) -> program 2 / 4
//
// Three things happen per element, all on the device, nothing allocated here:
//   fold        w[o, :] = g[o] * v[o, :] / ||v[o, :]||  (nn.utils.weight_norm, dim 0): k_row_rnorm (one block per row, ATen's summation
//               order) + two multiplies at gather time, in ATen's order (g * v) * (1 / ||v||);
//   permutation the blob's element j of chunk c reads source element src_of(c, j): CLOSED FORM - `chunk_desc` (what chunk c of a program
//               is) and `elem_src` (which weight its element j is) below are the library's statement of the layout the kernels consume
//               (mlp_chain.hip, mlp_bf16_core.h, mlp_k2_w32.hip, mlp_grad_bf16.hip, mlp_backward_bf16.hip).  Both are
//               __host__ __device__: the pack kernels evaluate them per element, and nerfart_pack_plan_debug evaluates the SAME functions on
//               the host into plain index tables, which tests/test_pack_plan.py holds equal - entry for entry, header word for header word -
//               to nerfart_amd/packing.py's numpy plans (the source of truth of the CPU emulation, tests/emul_chain.py);
//   split       split-bf16 / fp16 programs: hi = rne(w), lo = rne(w - hi), stored [unit][term][lane][4 x 2 halves].
#include "nerfart_common.h"
#include <hip/hip_fp16.h>

namespace nerfart {
namespace pack {

constexpr int KT = 16 * 64 * 4;       // indices per fp32 k tile
constexpr int KS = 16 * 64 * 8;       // indices per split k-step (16 output tiles)
constexpr int ENC = 39, HW = 217;     // encoding features; hidden width in front of the skip connection
constexpr int SRC_ZERO = -1, SRC_ONE = -2;

// ---- slot -> feature maps (packing.py: enc_slot_feature, unit_feature_*, w32_feature_*) -------------------------------------------
__host__ __device__ inline int enc_slot_feature(int slot) {
    const int t = slot / 16, rem = slot % 16, g = rem / 4, r = rem % 4, m = 4 * t + r;
    if (g < 3) {
        if (m == 0) return g;
        if (m == 11) return -1;
        const int k = (m - 1) / 2, is_cos = (m - 1) % 2;
        return 3 + 6 * k + 3 * is_cos + g;
    }
    if (m < 6) { const int i = m / 2, is_cos = m % 2; return 3 + 6 * 5 + 3 * is_cos + i; }
    return -1;
}
__host__ __device__ inline int unit_feature_hidden(int ks, int g, int e) { return 32 * ks + (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4)); }
__host__ __device__ inline int unit_feature_enc(int q, int g, int e) {
    const int m = 8 * q + e;
    if (g >= 3 || m > 12) return -1;
    if (m == 0) return g;
    const int k = (m - 1) / 2, is_cos = (m - 1) % 2;
    return 3 + 6 * k + 3 * is_cos + g;
}
__host__ __device__ inline int unit_feature_extra(int q, int g, int e, int n_extra) { const int m = 32 * q + 8 * g + e; return m < n_extra ? m : -1; }
__host__ __device__ inline int w32_feature_hidden(int ks, int h, int e) { return 16 * ks + 8 * (e >> 2) + 4 * h + (e & 3); }
__host__ __device__ inline int w32_feature_enc(int q, int h, int e) {
    int m = 8 * q + e;
    if (h == 0) { if (m < 3) return m; m -= 3; }
    if (m >= 18) return -1;
    const int kc = m / 2, is_cos = m % 2, k = kc / 3, c = kc % 3;
    return 3 + 6 * (k + 3 * h) + 3 * is_cos + c;
}

// ---- source tensors -----------------------------------------------------------------------------------------------------------------
// surface programs: tensor 2l = w_l, 2l + 1 = b_l (l = 0..8); radiance programs: 0 = w8, 1 = b8, 2 + 2l = r_l, 3 + 2l = rb_l (l = 0..4)
struct Dims { int rows, cols; };
__host__ __device__ inline Dims surf_dims(int l) { return Dims{l == 8 ? 257 : (l == 3 ? HW : 256), l == 0 ? ENC : 256}; }
__host__ __device__ inline Dims rad_dims(int l, int n_extra) { return l == 0 ? Dims{256, n_extra + 256} : (l == 4 ? Dims{3, 256} : Dims{256, 256}); }
__host__ __device__ inline Dims tensor_dims(bool radiance, int t, int n_extra) {      // (rows, cols) of weight tensor t (even ids; radiance: t = 0 is w8)
    if (!radiance) return surf_dims(t / 2);
    return t == 0 ? Dims{257, 256} : rad_dims((t - 2) / 2, n_extra);
}
struct Src { int t, row, col; };           // t >= 0: tensor t; a bias tensor uses row only; SRC_ZERO / SRC_ONE: the constants
__host__ __device__ inline Src zero_src() { return Src{SRC_ZERO, 0, 0}; }

// ---- chunk descriptors --------------------------------------------------------------------------------------------------------------
enum Kind { F32_FWD = 0, F32_REV = 1, F32_TAIL = 2, B16_FWD = 3, B16_T = 4, W32 = 5 };
enum ColMap { C_ENC = 0, C_SKIP = 1, C_HID = 2, C_RAD0 = 3 };
enum RowFeat { R_NAT = 0, R_ENC = 1, R_NRM = 2, R_RADFEAT = 3 };
struct Chunk {
    int kind, tensor, t0, nt;      // first k tile / k-step of the chunk and how many
    int colmap, limit;             // forward kinds: column map + its in_dim; transposed kinds: limit = number of valid k features
    int out_dim, row0;             // forward kinds: valid output rows, first source row; transposed: row0 added to the k feature
    int rowfeat, tile0, ntile;     // B16_T: which output rows, first output tile, tiles per k-step
    int mul;                       // B16_T: times w8[0][k feature] (the folded sdf row of the reverse-mode program)
    float scale;
    int n;                         // elements (indices) of the chunk
};

struct Prog { int prog, view_tiles, n_extra, fp16; bool radiance, split; };
__host__ __device__ inline Prog make_prog(int prog, int view_tiles, int fp16) {
    return Prog{prog, view_tiles, view_tiles == 1 ? 9 : 33, fp16, prog == 2 || prog == 4, prog >= 3};
}

__host__ __device__ inline int n_chunks(const Prog& p) {
    switch (p.prog) {
        case 1: return 59 + 59;
        case 2: return 8 + 8 + (p.view_tiles + 1) / 2 + 24;
        case 3: return 30 + 29 + 4 + 30;
        default: return 21 + 12 + 1 + 4 + 4;
    }
}

__host__ __device__ inline Chunk chunk_desc(const Prog& p, int c) {
    Chunk d{};
    d.scale = 1.f; d.ntile = 16; d.limit = 256; d.out_dim = 256;
    const float inv_unorm = (float)(1.0 / 65535.0);
    if (p.prog == 1) {
        if (c < 59) {                                   // forward: layer 0 (2 chunks), 1..3 (8 each), 4 (9), 5..7 (8 each)
            int l, k;
            if (c < 2) { l = 0; k = c; } else if (c < 26) { l = 1 + (c - 2) / 8; k = (c - 2) % 8; } else if (c < 35) { l = 4; k = c - 26; } else { l = 5 + (c - 35) / 8; k = (c - 35) % 8; }
            d.kind = F32_FWD; d.tensor = 2 * l; d.out_dim = surf_dims(l).rows; d.t0 = 2 * k;
            const int ntl = l == 0 ? 3 : (l == 4 ? 17 : 16);
            d.nt = (2 * k + 1 < ntl && !(l == 4 && k == 8)) ? 2 : 1;
            if (l == 4 && k == 8) d.t0 = 16;
            d.colmap = l == 0 ? C_ENC : (l == 4 ? C_SKIP : C_HID);
            d.n = d.nt * KT;
            return d;
        }
        c -= 59;                                        // reverse: 7, 6, 5 (8 each), tail of 4 (2), 4 (8), 3 (7), 2, 1 (8 each), tail of 0 (2)
        int l, k, tail = 0;
        if (c < 24) { l = 7 - c / 8; k = c % 8; } else if (c < 26) { l = 4; k = c - 24; tail = 1; } else if (c < 34) { l = 4; k = c - 26; }
        else if (c < 41) { l = 3; k = c - 34; } else if (c < 57) { l = 2 - (c - 41) / 8; k = (c - 41) % 8; } else { l = 0; k = c - 57; tail = 1; }
        d.tensor = 2 * l; d.limit = surf_dims(l).rows;
        if (tail) { d.kind = F32_TAIL; d.t0 = 8 * k; d.nt = 8; d.ntile = 3; d.colmap = l == 4 ? C_SKIP : C_ENC; d.n = d.nt * 3 * 64 * 4; }
        else { d.kind = F32_REV; d.t0 = 2 * k; d.nt = 2; d.colmap = l == 4 ? C_SKIP : C_HID; d.n = 2 * KT; }
        return d;
    }
    if (p.prog == 2) {
        const int ne_chunks = (p.view_tiles + 1) / 2;
        d.kind = F32_FWD; d.colmap = C_HID;
        if (c < 8) { d.tensor = 0; d.row0 = 1; d.t0 = 2 * c; d.nt = 2; }
        else if (c < 16 + ne_chunks) {
            const int k = c - 8;
            d.tensor = 2; d.colmap = C_RAD0; d.t0 = 2 * k;
            d.nt = k < 8 ? 2 : ((16 + p.view_tiles) - 2 * k >= 2 ? 2 : 1);
        } else { const int k = c - 16 - ne_chunks; d.tensor = 4 + 2 * (k / 8); d.t0 = 2 * (k % 8); d.nt = 2; }
        d.n = d.nt * KT;
        return d;
    }
    if (p.prog == 3) {
        if (c < 30 || c >= 63) {                        // forward layers: 16x16x32 fragments (chunks 0..29) / 32x32x16 fragments (63..92)
            const bool w32 = c >= 63;
            if (w32) c -= 63;
            int l, k;
            if (c < 1) { l = 0; k = 0; } else if (c < 13) { l = 1 + (c - 1) / 4; k = (c - 1) % 4; } else if (c < 18) { l = 4; k = c - 13; } else { l = 5 + (c - 18) / 4; k = (c - 18) % 4; }
            d.tensor = 2 * l; d.out_dim = surf_dims(l).rows; d.limit = surf_dims(l).cols;
            d.colmap = l == 0 ? C_ENC : (l == 4 ? C_SKIP : C_HID);
            if (w32) {
                const int nks = l == 0 ? 3 : (l == 4 ? 17 : 16);
                d.kind = W32; d.t0 = 4 * k; d.nt = nks - 4 * k < 4 ? nks - 4 * k : 4; d.n = d.nt * 8 * 512;
            } else {
                d.kind = B16_FWD; d.t0 = 2 * k; d.nt = l == 0 ? 2 : ((l == 4 && k == 4) ? 1 : 2); d.n = d.nt * KS;
            }
            return d;                                   // (the skip layer's 1 / sqrt 2 is a TENSOR scale: value_of)
        }
        d.kind = B16_T; d.rowfeat = R_NAT;
        if (c < 58) {                                   // reverse-mode program: layers 7..1, 4 chunks each
            const int l = 7 - (c - 30) / 4, k = (c - 30) % 4;
            d.tensor = 2 * l; d.limit = surf_dims(l).rows; d.t0 = 2 * k; d.nt = 2; d.n = 2 * KS;
            d.mul = l == 7; d.scale = (l == 7 || p.fp16) ? 1.f : inv_unorm;
        } else if (c == 58) {                           // the 39 encoding rows of layer 0 at rows 217..255: 3 output tiles, all 8 k-steps
            d.tensor = 0; d.limit = 256; d.t0 = 0; d.nt = 8; d.rowfeat = R_ENC; d.tile0 = 13; d.ntile = 3; d.n = 8 * 3 * 512;
            d.scale = p.fp16 ? 1.f : inv_unorm;
        } else {                                        // second-order program: layer 7 without the folded sdf row
            d.tensor = 14; d.limit = 256; d.t0 = 2 * (c - 59); d.nt = 2; d.n = 2 * KS; d.scale = inv_unorm;
        }
        return d;
    }
    // prog 4
    const int ex = p.view_tiles == 1 ? 1 : 2;
    if (c < 21) {
        d.kind = B16_FWD; d.colmap = C_HID; d.nt = 2;
        if (c < 4) { d.tensor = 0; d.row0 = 1; d.t0 = 2 * c; }
        else if (c < 9) { d.tensor = 2; d.colmap = C_RAD0; d.t0 = 2 * (c - 4); if (c == 8) d.nt = ex; }
        else { const int k = c - 9; d.tensor = 4 + 2 * (k / 4); d.t0 = 2 * (k % 4); }
        d.n = d.nt * KS;
        return d;
    }
    d.kind = B16_T; d.rowfeat = R_NAT; d.nt = 2; d.n = 2 * KS;
    c -= 21;
    if (c < 12) { d.tensor = 8 - 2 * (c / 4); d.t0 = 2 * (c % 4); }                         // r3, r2, r1
    else if (c == 12) { d.tensor = 2; d.t0 = 0; d.nt = 8; d.rowfeat = R_NRM; d.tile0 = 0; d.ntile = 1; d.n = 8 * 512; }   // normal rows of r0^T
    else if (c < 17) { d.tensor = 2; d.t0 = 2 * (c - 13); d.rowfeat = R_RADFEAT; }         // feature rows of r0^T
    else { d.tensor = 0; d.row0 = 1; d.t0 = 2 * (c - 17); }                                // w8[1:]^T
    return d;
}

// column slot -> source column of a forward layer (fp32 programs: 16-wide slots; -1 = zero)
__host__ __device__ inline int f32_col(const Prog& p, int colmap, int slot) {
    switch (colmap) {
        case C_ENC: return enc_slot_feature(slot);
        case C_SKIP: if (slot < 224) return slot < HW ? slot : -1; { const int e = enc_slot_feature(slot - 224); return e >= 0 ? HW + e : -1; }
        case C_RAD0: if (slot < 256) return p.n_extra + slot; return slot - 256 < p.n_extra ? slot - 256 : -1;
        default: return slot < 256 ? slot : -1;
    }
}
__host__ __device__ inline int b16_col(const Prog& p, const Chunk& d, int ks, int g, int e) {
    switch (d.colmap) {
        case C_ENC: return unit_feature_enc(ks, g, e);
        case C_SKIP: { if (ks < 7) { const int f = unit_feature_hidden(ks, g, e); return f < HW ? f : -1; } const int f = unit_feature_enc(ks - 7, g, e); return f >= 0 ? HW + f : -1; }
        case C_RAD0: return ks < 8 ? p.n_extra + unit_feature_hidden(ks, g, e) : unit_feature_extra(ks - 8, g, e, p.n_extra);
        default: { const int f = unit_feature_hidden(ks, g, e); return f < d.limit ? f : -1; }
    }
}
__host__ __device__ inline int w32_col(const Chunk& d, int ks, int h, int e) {
    switch (d.colmap) {
        case C_ENC: return w32_feature_enc(ks, h, e);
        case C_SKIP: { if (ks < 14) { const int f = w32_feature_hidden(ks, h, e); return f < HW ? f : -1; } const int f = w32_feature_enc(ks - 14, h, e); return f >= 0 ? HW + f : -1; }
        default: { const int f = w32_feature_hidden(ks, h, e); return f < d.limit ? f : -1; }
    }
}

// element j of chunk d -> its source (and, for B16_T chunks with d.mul, the second factor w8[0][k feature] in `mul`)
__host__ __device__ inline Src elem_src(const Prog& p, const Chunk& d, int j, Src* mul) {
    if (mul) *mul = Src{SRC_ONE, 0, 0};
    switch (d.kind) {
        case F32_FWD: {
            const int tl = j / KT, T = (j / 256) % 16, lane = (j / 4) % 64, r = j % 4, g = lane / 16, i = lane % 16;
            const int row = 16 * T + i, col = f32_col(p, d.colmap, 16 * (d.t0 + tl) + 4 * g + r);
            if (row >= d.out_dim || col < 0) return zero_src();
            return Src{d.tensor, d.row0 + row, col};
        }
        case F32_REV: case F32_TAIL: {
            const int per = d.ntile * 256, tl = j / per, T = (j / 256) % d.ntile, lane = (j / 4) % 64, r = j % 4, g = lane / 16, i = lane % 16;
            const int kf = 16 * (d.t0 + tl) + 4 * g + r, o = 16 * T + i;
            int of;
            if (d.kind == F32_TAIL) { const int e = enc_slot_feature(o); of = d.colmap == C_SKIP ? (e >= 0 ? HW + e : -1) : e; }
            else of = d.colmap == C_SKIP ? (o < HW ? o : -1) : o;
            if (kf >= d.limit || of < 0) return zero_src();
            return Src{d.tensor, kf, of};
        }
        case B16_FWD: {
            const int ksl = j / KS, T = (j / 512) % 16, lane = (j / 8) % 64, e = j % 8, g = lane / 16, i = lane % 16;
            const int row = 16 * T + i, col = b16_col(p, d, d.t0 + ksl, g, e);
            if (row >= d.out_dim || col < 0) return zero_src();
            return Src{d.tensor, d.row0 + row, col};
        }
        case B16_T: {
            const int per = d.ntile * 512, ksl = j / per, n = (j / 512) % d.ntile, lane = (j / 8) % 64, e = j % 8, g = lane / 16, i = lane % 16;
            const int o = 16 * (d.tile0 + n) + i;
            int kf = unit_feature_hidden(d.t0 + ksl, g, e);
            if (kf >= d.limit) kf = -1;
            if (mul && d.mul && kf >= 0 && kf < 256) *mul = Src{16, 0, kf};
            int rf;
            switch (d.rowfeat) {
                case R_ENC: rf = o >= HW ? o - HW : -1; break;
                case R_NRM: rf = o < 3 ? p.n_extra - 3 + o : -1; break;
                case R_RADFEAT: rf = p.n_extra + o; break;
                default: rf = o;
            }
            const Dims dm = tensor_dims(p.radiance, d.tensor, p.n_extra);
            if (rf < 0 || kf < 0 || kf + d.row0 >= dm.rows) return zero_src();
            return Src{d.tensor, kf + d.row0, rf};
        }
        default: {   // W32
            const int item = j / 512, ksl = item / 8, T = item % 8, lane = (j / 8) % 64, e = j % 8;
            const int row = 32 * T + (lane & 31), col = w32_col(d, d.t0 + ksl, lane >> 5, e);
            if (row >= d.out_dim || col < 0) return zero_src();
            return Src{d.tensor, row, col};
        }
    }
}

// aux element j (biases, the rows the VALU epilogues read)
__host__ __device__ inline int aux_len(const Prog& p) { return p.radiance ? 2052 : 2308; }
__host__ __device__ inline Src aux_src(const Prog& p, int j) {
    if (!p.radiance) {
        if (j < 2048) { const int l = j / 256, n = j % 256; return n < surf_dims(l).rows ? Src{2 * l + 1, n, 0} : zero_src(); }
        if (j < 2304) return Src{16, 0, j - 2048};
        return j == 2304 ? Src{17, 0, 0} : zero_src();
    }
    if (j < 256) return Src{1, 1 + j, 0};
    if (j < 1280) return Src{3 + 2 * ((j - 256) / 256), (j - 256) % 256, 0};
    if (j < 2048) return Src{10, (j - 1280) / 256, (j - 1280) % 256};
    return j - 2048 < 3 ? Src{11, j - 2048, 0} : zero_src();
}

// ---- header ---------------------------------------------------------------------------------------------------------------------------
struct Layout { int nc_total, nc, nc_all, aux_off, total, pad, offs[128]; };
__host__ inline Layout make_layout(const Prog& p) {
    Layout L{};
    L.nc_total = n_chunks(p);
    L.offs[0] = NERFART_HDR_INTS;
    for (int c = 0; c < L.nc_total; ++c) L.offs[c + 1] = L.offs[c] + chunk_desc(p, c).n;
    L.aux_off = L.offs[L.nc_total];
    switch (p.prog) {
        case 1: L.nc = 59; L.nc_all = 118; break;
        case 2: L.nc = L.nc_total; L.nc_all = L.nc_total; break;
        case 3: L.nc = 30; L.nc_all = 59; break;
        default: L.nc = 21; L.nc_all = L.nc_total;
    }
    const int body_end = L.aux_off + aux_len(p);
    L.total = body_end;
    if (p.split) { const int need = L.offs[L.nc_total - 1] + 2 * 16 * 512; if (need > L.total) L.total = need; }   // the weight stream always copies 64 KiB
    L.pad = L.total - body_end;
    return L;
}
__host__ inline void make_header(const Prog& p, const Layout& L, int* hdr) {
    for (int i = 0; i < NERFART_HDR_INTS; ++i) hdr[i] = 0;
    hdr[0] = NERFART_MAGIC; hdr[1] = p.prog; hdr[2] = L.nc; hdr[3] = L.total; hdr[4] = L.aux_off; hdr[5] = aux_len(p); hdr[6] = L.nc_all;
    for (int c = 0; c <= L.nc_total; ++c) hdr[NERFART_HDR_OFFS + c] = L.offs[c];
    if (p.prog == 3) {
        hdr[7] = 28;                                     // the second-order program: B7' (chunks 59..62), then B6..B1 (34..57)
        for (int i = 0; i < 4; ++i) hdr[128 + i] = L.offs[59 + i];
        for (int i = 0; i < 24; ++i) hdr[132 + i] = L.offs[34 + i];
        hdr[8] = 30; hdr[9] = 63;                        // the 32x32x16-fragment program: chunk count, first chunk
    }
    // fragment encoding: 1 = bf16 hi + lo, 2 = fp16 hi + lo (ABI 3), 3 = fp16 hi + lo of a SOFTPLUS-SCALED network (ABI 5, precision 5: the caller's tensors
    // carry the scale, see nerfart_pack_surface_blob) - same layout as 2, read by the 1-MFMA K2 only
    if (p.split) hdr[10] = p.fp16 ? (p.fp16 == 2 ? 3 : 2) : 1;
}

// ---- device side ----------------------------------------------------------------------------------------------------------------------
struct Tensors {
    const float* g[9]; const float* v[9]; const float* b[9];   // per LAYER (surface: l = 0..8; radiance: 0 = the surface's layer 8, 1 + l = r_l)
    int rn_off[9];                                             // offset of the layer's 1 / ||v|| row vector in the workspace
};
struct PackArgs { Prog p; Tensors t; const float* rnorm; float* blob; int nc_total, aux_off, total, offs[128]; int hdr_small[16]; int table2[32]; };

__device__ __forceinline__ float value_of(const PackArgs& a, Src s) {
    if (s.t == SRC_ZERO) return 0.f;
    if (s.t == SRC_ONE) return 1.f;
    const int l = s.t >> 1;
    if (s.t & 1) return a.t.b[l][s.row];
    const int cols = tensor_dims(a.p.radiance, s.t, a.p.n_extra).cols;
    float w = (a.t.g[l][s.row] * a.t.v[l][(size_t)s.row * cols + s.col]) * a.rnorm[a.t.rn_off[l] + s.row];   // ATen's (g * v) * (1 / ||v||)
    if (a.p.prog == 3 && s.t == 8) w *= 0.70710678118654752440f;          // cat[h, enc] / sqrt 2 (base.py:250) folded into the skip layer's weights
    return w;
}

// 1 / ||v[row, :]|| for every row of every layer: one 256-thread block per row, summed in the order of ATen's weight_norm forward
// (aten/src/ATen/native/cuda/WeightNorm.cu, weight_norm_fwd_first_dim_kernel + reduce_block_into_lanes: thread t accumulates columns
// t, t + 256, ...; shared-memory halving 128, 64; x[t] + x[t + 32]; shuffle-down 16 .. 1; sqrtf; 1.f / result), so that the folded weights
// - and with them every blob - equal torch._weight_norm's on the same device bit for bit (tests/test_gpu_pack.py)
__global__ void __launch_bounds__(256) k_row_rnorm(Tensors t, Prog p, int n_layers, float* __restrict__ rnorm) {
    __shared__ float x[256];
    const int tid = threadIdx.x;
    int l = 0, row = blockIdx.x;
    for (; l < n_layers; ++l) {
        const int rows = p.radiance ? (l == 0 ? 257 : rad_dims(l - 1, p.n_extra).rows) : surf_dims(l).rows;
        if (row < rows) break;
        row -= rows;
    }
    if (l >= n_layers) return;
    const int cols = p.radiance ? (l == 0 ? 256 : rad_dims(l - 1, p.n_extra).cols) : surf_dims(l).cols;
    const float* v = t.v[l] + (size_t)row * cols;
    float thread_sum = 0.f;
    for (int c = tid; c < cols; c += 256) { const float val = v[c]; thread_sum += val * val; }
    x[tid] = thread_sum;
    __syncthreads();
    if (tid < 128) x[tid] = x[tid] + x[tid + 128];
    __syncthreads();
    if (tid < 64) x[tid] = x[tid] + x[tid + 64];
    __syncthreads();
    if (tid < 32) {
        float fin = x[tid] + x[tid + 32];
        for (int i = 16; i >= 1; i >>= 1) fin = fin + __shfl_down(fin, i);
        if (tid == 0) { const float result = sqrtf(fin); rnorm[t.rn_off[l] + row] = 1.f / result; }
    }
}

__device__ __forceinline__ int find_chunk(const PackArgs& a, int off) {
    int lo = 0, hi = a.nc_total - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (a.offs[mid] <= off) lo = mid; else hi = mid - 1; }
    return lo;
}
__device__ __forceinline__ unsigned short bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// header + chunks + aux + padding of one blob; one thread per output float (split programs: per (hi, lo) pair of floats)
__global__ void __launch_bounds__(256) k_pack(PackArgs a) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < NERFART_HDR_INTS) {
        int w = 0;
        if (tid < 16) w = a.hdr_small[tid];
        else if (tid <= NERFART_HDR_OFFS + a.nc_total) w = a.offs[tid - NERFART_HDR_OFFS];
        else if (a.p.prog == 3 && tid >= 128 && tid < 128 + 28) w = a.table2[tid - 128];
        a.blob[tid] = __int_as_float(w);
        return;
    }
    const int body = a.aux_off - NERFART_HDR_INTS;
    int q = tid - NERFART_HDR_INTS;
    if (!a.p.split) {
        if (q < body) {
            const int off = NERFART_HDR_INTS + q, c = find_chunk(a, off);
            const Chunk d = chunk_desc(a.p, c);
            a.blob[off] = value_of(a, elem_src(a.p, d, off - a.offs[c], nullptr));
            return;
        }
        q -= body;
    } else {
        if (q < body / 2) {                              // unit = 512 floats: [term][lane][4]; this thread: (unit, lane, pair)
            const int unit = q / 256, lane = (q / 4) % 64, pr = q % 4;
            const int off = NERFART_HDR_INTS + unit * 512, c = find_chunk(a, off);
            const Chunk d = chunk_desc(a.p, c);
            const int j0 = (off - a.offs[c]) + lane * 8 + 2 * pr;
            unsigned hi_bits = 0, lo_bits = 0;
            for (int x = 0; x < 2; ++x) {
                Src m;
                float w = value_of(a, elem_src(a.p, d, j0 + x, &m));
                if (d.mul) w = w * value_of(a, m);
                if (d.scale != 1.f) w = w * d.scale;
                unsigned short h, l;
                if (a.p.fp16) {
                    const __half hh = __float2half_rn(w);
                    const __half ll = __float2half_rn(w - __half2float(hh));
                    h = __half_as_ushort(hh); l = __half_as_ushort(ll);
                } else {
                    h = bf16_rne(w); l = bf16_rne(w - bf16_to_f(h));
                }
                hi_bits |= (unsigned)h << (16 * x); lo_bits |= (unsigned)l << (16 * x);
            }
            a.blob[off + lane * 4 + pr] = __uint_as_float(hi_bits);
            a.blob[off + 256 + lane * 4 + pr] = __uint_as_float(lo_bits);
            return;
        }
        q -= body / 2;
    }
    const int n_aux = aux_len(a.p);
    if (q < n_aux) { a.blob[a.aux_off + q] = value_of(a, aux_src(a.p, q)); return; }
    q -= n_aux;
    if (a.aux_off + n_aux + q < a.total) a.blob[a.aux_off + n_aux + q] = 0.f;
}

static int launch_pack(const Prog& p, const Tensors& t, int n_layers, float* blob, long long blob_floats, void* workspace, long long workspace_bytes,
                       hipStream_t stream) {
    const Layout L = make_layout(p);
    if (!blob || blob_floats < L.total) { set_last_error("pack: blob_out is smaller than nerfart_blob_floats()"); return 2; }
    int rows = 0;
    Tensors tt = t;
    for (int l = 0; l < n_layers; ++l) {
        if (!t.g[l] || !t.v[l] || !t.b[l]) { set_last_error("pack: NULL weight_g / weight_v / bias pointer"); return 2; }
        tt.rn_off[l] = rows;
        rows += p.radiance ? (l == 0 ? 257 : rad_dims(l - 1, p.n_extra).rows) : surf_dims(l).rows;
    }
    if (!workspace || workspace_bytes < (long long)rows * 4) { set_last_error("pack: workspace smaller than nerfart_pack_workspace_bytes()"); return 2; }
    float* rnorm = (float*)workspace;
    hipLaunchKernelGGL(k_row_rnorm, dim3(rows), dim3(256), 0, stream, tt, p, n_layers, rnorm);
    NERFART_HIP(hipGetLastError());
    PackArgs a{};
    a.p = p; a.t = tt; a.rnorm = rnorm; a.blob = blob; a.nc_total = L.nc_total; a.aux_off = L.aux_off; a.total = L.total;
    for (int c = 0; c <= L.nc_total; ++c) a.offs[c] = L.offs[c];
    int hdr[NERFART_HDR_INTS];
    make_header(p, L, hdr);
    for (int i = 0; i < 16; ++i) a.hdr_small[i] = hdr[i];
    for (int i = 0; i < 28; ++i) a.table2[i] = hdr[128 + i];
    const int body = L.aux_off - NERFART_HDR_INTS;
    const long long threads = NERFART_HDR_INTS + (p.split ? body / 2 : body) + aux_len(p) + L.pad;
    hipLaunchKernelGGL(k_pack, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, a);
    NERFART_HIP(hipGetLastError());
    blob_term_register(blob, hdr[10]);        // what the entry points that read one fragment encoding check this pointer against (capi_common.cpp)
    return 0;
}

static int prog_of(int precision, bool radiance, int* fp16) {
    *fp16 = precision == 4 ? 1 : (precision == 5 ? 2 : 0);
    if (precision == 0) return radiance ? 2 : 1;
    if (precision == 1 || precision == 4) return radiance ? 4 : 3;
    if (precision == 5 && !radiance) return 3;            // the SDF net only: Algorithm 1's sampler (csrc/mlp_chain_f16x1.hip)
    return 0;
}

}  // namespace pack
}  // namespace nerfart

using namespace nerfart;
using namespace nerfart::pack;

extern "C" {

// floats of the blob of (precision, which net): precision 0 fp32, 1 split bf16, 4 fp16 hi + lo (C-ABI precision ids); 0 on a bad argument
long long nerfart_surface_blob_floats(int precision, int multires) {
    int f; const int prog = prog_of(precision, false, &f);
    if (!prog || multires != 6) { set_last_error("surface blob: precision must be 0, 1 or 4 and multires 6 (the shipped configs)"); return 0; }
    return make_layout(make_prog(prog, 1, f)).total;
}
long long nerfart_radiance_blob_floats(int precision, int view_tiles) {
    int f; const int prog = prog_of(precision, true, &f);
    if (!prog || (view_tiles != 1 && view_tiles != 3)) { set_last_error("radiance blob: precision must be 0, 1 or 4 and view_tiles 1 or 3"); return 0; }
    return make_layout(make_prog(prog, view_tiles, f)).total;
}
long long nerfart_pack_workspace_bytes(void) { return 4096 * 4; }

// The architecture the packers (and every kernel) are written for, as data a caller can check its tensors against BEFORE handing over pointer
// tables the library indexes with fixed dims: radiance == 0: the SDF net's 9 layers for embed_multires = arg; != 0: the radiance net's 5 layers for
// view_tiles = arg.  rows[l] / cols[l] = weight_v[l].shape (weight_g [rows, 1], bias [rows]).  Returns the layer count; 0 on a bad argument.
int nerfart_pack_layer_dims(int radiance, int arg, int* rows, int* cols) {
    if (!radiance) {
        if (arg != 6) { set_last_error("pack_layer_dims: the SDF kernels are written for embed_multires 6"); return 0; }
        for (int l = 0; l < 9; ++l) { const Dims d = surf_dims(l); if (rows) rows[l] = d.rows; if (cols) cols[l] = d.cols; }
        return 9;
    }
    if (arg != 1 && arg != 3) { set_last_error("pack_layer_dims: view_tiles must be 1 (raw view directions) or 3 (embed_multires_view 4)"); return 0; }
    const int n_extra = make_prog(2, arg, 0).n_extra;
    for (int l = 0; l < 5; ++l) { const Dims d = rad_dims(l, n_extra); if (rows) rows[l] = d.rows; if (cols) cols[l] = d.cols; }
    return 5;
}

// weight_g[l] [out_l] (or [out_l, 1]), weight_v[l] [out_l, in_l], bias[l] [out_l]: HOST arrays of 9 DEVICE pointers, the state-dict tensors
// `implicit_surface.surface_fc_layers.{l}.weight_g / weight_v / bias` (W = 256, D = 8, skips = [4], embed_multires = 6, W_geo_feat = 256).
int nerfart_pack_surface_blob(int precision, int multires, const float* const* weight_g, const float* const* weight_v, const float* const* bias,
                              float* blob_out, long long blob_floats, void* workspace, long long workspace_bytes, void* stream) {
    int f; const int prog = prog_of(precision, false, &f);
    if (!prog || multires != 6) { set_last_error("pack_surface_blob: precision must be 0, 1 or 4 and multires 6"); return 2; }
    if (!weight_g || !weight_v || !bias) { set_last_error("pack_surface_blob: NULL pointer table"); return 2; }
    Tensors t{};
    for (int l = 0; l < 9; ++l) { t.g[l] = weight_g[l]; t.v[l] = weight_v[l]; t.b[l] = bias[l]; }
    return launch_pack(make_prog(prog, 1, f), t, 9, blob_out, blob_floats, workspace, workspace_bytes, (hipStream_t)stream);
}

// surf8_*: the SDF net's LAST layer (`surface_fc_layers.8`: rows 1..256 produce the geometry feature the radiance kernels evaluate);
// weight_g / weight_v / bias: HOST arrays of 5 device pointers, `radiance_net.layers.{l}.*` (W = 256, D = 4; input [x, v(, embedded), n, feat]).
int nerfart_pack_radiance_blob(int precision, int view_tiles, const float* surf8_g, const float* surf8_v, const float* surf8_bias,
                               const float* const* weight_g, const float* const* weight_v, const float* const* bias, float* blob_out, long long blob_floats,
                               void* workspace, long long workspace_bytes, void* stream) {
    int f; const int prog = prog_of(precision, true, &f);
    if (!prog || (view_tiles != 1 && view_tiles != 3)) { set_last_error("pack_radiance_blob: precision must be 0, 1 or 4 and view_tiles 1 or 3"); return 2; }
    if (!weight_g || !weight_v || !bias) { set_last_error("pack_radiance_blob: NULL pointer table"); return 2; }
    Tensors t{};
    t.g[0] = surf8_g; t.v[0] = surf8_v; t.b[0] = surf8_bias;
    for (int l = 0; l < 5; ++l) { t.g[1 + l] = weight_g[l]; t.v[1 + l] = weight_v[l]; t.b[1 + l] = bias[l]; }
    return launch_pack(make_prog(prog, view_tiles, f), t, 6, blob_out, blob_floats, workspace, workspace_bytes, (hipStream_t)stream);
}

// HOST-ONLY, for tests: the plan of a program as plain tables, produced by the functions the pack kernel evaluates.
//   sizes[0] = chunk elements, sizes[1] = aux elements, sizes[2] = total floats, sizes[3] = chunks; pass NULL tables to query sizes only.
//   cindex / cmul: per chunk element, the FLAT source index in packing.py's `_Flat` order (tensors concatenated, then 0.0, then 1.0);
//   cscale: per chunk element; aindex: per aux element; header: the 512 header words.
int nerfart_pack_plan_debug(int program, int view_tiles, int fp16, long long* sizes, int* header, int* cindex, int* cmul, float* cscale, int* aindex) {
    if (program < 1 || program > 4 || (view_tiles != 1 && view_tiles != 3)) { set_last_error("pack_plan_debug: program 1..4, view_tiles 1 or 3"); return 2; }
    const Prog p = make_prog(program, view_tiles, fp16);
    const Layout L = make_layout(p);
    const int nt = p.radiance ? 12 : 18;
    long long base[19]; long long n = 0;
    for (int t = 0; t < nt; ++t) {
        base[t] = n;
        const Dims d = tensor_dims(p.radiance, t & ~1, p.n_extra);
        n += (t & 1) ? d.rows : (long long)d.rows * d.cols;
    }
    auto flat = [&](Src s) -> int {
        if (s.t == SRC_ZERO) return (int)n;
        if (s.t == SRC_ONE) return (int)n + 1;
        if (s.t & 1) return (int)(base[s.t] + s.row);
        return (int)(base[s.t] + (long long)s.row * tensor_dims(p.radiance, s.t, p.n_extra).cols + s.col);
    };
    if (sizes) { sizes[0] = L.aux_off - NERFART_HDR_INTS; sizes[1] = aux_len(p); sizes[2] = L.total; sizes[3] = L.nc_total; }
    if (header) make_header(p, L, header);
    if (cindex) {
        long long k = 0;
        for (int c = 0; c < L.nc_total; ++c) {
            const Chunk d = chunk_desc(p, c);
            for (int j = 0; j < d.n; ++j, ++k) {
                Src m; const Src s = elem_src(p, d, j, &m);
                cindex[k] = flat(s);
                if (cmul) cmul[k] = flat(m);
                if (cscale) cscale[k] = d.scale;
            }
        }
    }
    if (aindex) for (int j = 0; j < aux_len(p); ++j) aindex[j] = flat(aux_src(p, j));
    return 0;
}

}  // extern "C"
