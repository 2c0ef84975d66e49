// wgrad.hip - weight-gradient reductions of the fine-tune step's pass 2 (row a19; reference: autograd through
// models/frameworks/volsdf.py:759-770 accumulates dW_l = sum over points of delta_l^T act_{l-1} and db_l = sum delta_l).
//
//   dW[mat][p][q] = sum_r Z_mat[r][p] * A_mat[r][q]          r over `rows` dump rows (points, or points + tangent rows)
//   cs[mat][p]    = sum_{r < cs_rows} Z_mat[r][p]            (the bias gradients: column sums of the deltas)
//
// Z_mat [rows, 256] and A_mat [rows, NA] (NA = 256 or 64) are the bf16 POINT-MAJOR dumps the backward kernels wrote
// (mlp_backward_bf16.hip; include/nerfart_hip.h "dumps"), read in place.  Both MFMA operands are k-strided in memory (k = the row),
// so a 32-row tile is staged global -> registers -> LDS as [4-row][16-column] 128-byte subtiles and read back with
// ds_read_b64_tr_b16 (the hardware 4 x 4 transposing read): lane (column c of a 16-lane group) receives 4 consecutive rows of its
// column, two reads = the 8 k-values of a v_mfma_f32_32x32x16_bf16 operand.  The four lane groups of a read cover 512 contiguous
// bytes: conflict free.  One workgroup (8 waves) owns the whole 256 x NA fp32 result in registers (128 accumulator VGPRs per wave at
// NA = 256) and walks its slice of the rows; the row range is split across workgroups (split-K), partial results go to the
// caller's workspace and k_wgrad_reduce adds them up.  The column sums ride along in the staging registers (each thread always
// stages the same 8 columns), so the dumps are read ONCE - round 2 ran hipBLASLt batched GEMMs plus separate ATen column-sum
// reductions over the same bytes.
//
// Bound: HBM.  Algorithmic bytes per row and matrix pair: 2 * (256 + NA) (bf16); 2 * 256 * NA flop per row: 128 flop / byte at
// NA = 256 against a machine balance of ~400 - the matrix pipe runs at a third of its rate when the dumps stream at 6 TB/s.
#include "mlp_common.h"

namespace nerfart {
namespace wgrad {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int KT = 32;                 // rows per stage
constexpr int THREADS = 512;

template <int NA> struct Shape;
template <> struct Shape<256> { static constexpr int MB = 4, NB = 2; };     // per wave: 4 x 2 tiles of 32 x 32 (waves 2 x 4)
template <> struct Shape<64> { static constexpr int MB = 1, NB = 2; };      // per wave: 1 x 2 tiles (waves 8 x 1)

struct Args {
    const char* Z; long long z_stride;          // bytes between consecutive matrices
    const char* A; long long a_stride;
    long long rows, cs_rows;
    int n_split;                                // workgroups per matrix
    float* part;                                // [n_mats][n_split][256 * NA + 256]
};

// two transposing reads = one MFMA operand (8 bf16: rows 4 kg .. 4 kg + 7 of this lane's column).  ISSUED only: the caller waits once for a whole
// k-step's operands (frags_wait) - round 6: the per-fragment `s_waitcnt lgkmcnt(0)` of rounds 3-5 serialised 12 LDS latencies per stage in front of
// the MFMAs (mfma_util 0.31); now a k-step's 12 reads are in flight together and the NEXT k-step's are issued under this one's 8 MFMAs
struct Frag { u32x2 a, b; };
template <int O0, int O1>
__device__ __forceinline__ void frag_issue(Frag& f, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                 : "=&v"(f.a), "=&v"(f.b) : "v"(addr), "i"(O0), "i"(O1) : "memory");
}
__device__ __forceinline__ bf16x8 frag_value(const Frag& f) {
    u32x4 r = {f.a[0], f.a[1], f.b[0], f.b[1]};
    return __builtin_bit_cast(bf16x8, r);
}

template <int NA> struct Frags { Frag z[Shape<NA>::MB], a[Shape<NA>::NB]; };

// k-step S (16 rows) of a staged tile: fragment of column-block pair mb = subtile rows 4 S + {0, 1} (+ 2 for the upper lane half,
// which sits in the lane's base address), column blocks 2 mb + {0, 1} (the odd one for lanes 16..31 / 48..63, in the base too)
template <int NA, int S>
__device__ __forceinline__ void frags_issue(Frags<NA>& f, unsigned za, unsigned aa) {
    constexpr int MB = Shape<NA>::MB, NZ = 16, NAB = NA / 16;
    frag_issue<(0 + (4 * S) * NZ) * 128, (0 + (4 * S + 1) * NZ) * 128>(f.z[0], za);
    if constexpr (MB > 1) {
        frag_issue<(2 + (4 * S) * NZ) * 128, (2 + (4 * S + 1) * NZ) * 128>(f.z[1], za);
        frag_issue<(4 + (4 * S) * NZ) * 128, (4 + (4 * S + 1) * NZ) * 128>(f.z[2], za);
        frag_issue<(6 + (4 * S) * NZ) * 128, (6 + (4 * S + 1) * NZ) * 128>(f.z[3], za);
    }
    frag_issue<(0 + (4 * S) * NAB) * 128, (0 + (4 * S + 1) * NAB) * 128>(f.a[0], aa);
    frag_issue<(2 + (4 * S) * NAB) * 128, (2 + (4 * S + 1) * NAB) * 128>(f.a[1], aa);
}
// every LDS read issued so far has landed; the registers of `f` are named as in-outs so that no use of them is scheduled above the wait
template <int NA>
__device__ __forceinline__ void frags_wait(Frags<NA>& f) {
    constexpr int MB = Shape<NA>::MB;
    if constexpr (MB > 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.z[0].a), "+v"(f.z[0].b), "+v"(f.z[1].a), "+v"(f.z[1].b), "+v"(f.z[2].a), "+v"(f.z[2].b), "+v"(f.z[3].a),
                     "+v"(f.z[3].b), "+v"(f.a[0].a), "+v"(f.a[0].b), "+v"(f.a[1].a), "+v"(f.a[1].b) :: "memory");
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.z[0].a), "+v"(f.z[0].b), "+v"(f.a[0].a), "+v"(f.a[0].b), "+v"(f.a[1].a), "+v"(f.a[1].b) :: "memory");
    }
}
template <int NA>
__device__ __forceinline__ void frags_mma(const Frags<NA>& f, f32x16 (&acc)[Shape<NA>::MB][Shape<NA>::NB]) {
    constexpr int MB = Shape<NA>::MB, NB = Shape<NA>::NB;
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_value(f.z[a]), frag_value(f.a[b]), acc[a][b], 0, 0, 0);
}

template <int NA>
__global__ __launch_bounds__(THREADS) void k_wgrad(Args g) {
    constexpr int MB = Shape<NA>::MB, NB = Shape<NA>::NB;
    constexpr int NCB_Z = 16, NCB_A = NA / 16;                  // 16-column blocks per row
    constexpr int ZT = KT * 512, AT = KT * NA * 2;               // tile bytes
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Zs = smem;                                            // [2][ZT]
    char* As = smem + 2 * ZT;                                   // [2][AT]
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int mat = blockIdx.y, sp = blockIdx.x;
    const char* Z = g.Z + (size_t)mat * g.z_stride;
    const char* A = g.A + (size_t)mat * g.a_stride;
    // this workgroup's rows: [r_begin, r_end), boundaries on stage multiples
    const long long stages = (g.rows + KT - 1) / KT;
    const long long per = (stages + g.n_split - 1) / g.n_split;
    const long long r_begin = (long long)sp * per * KT;
    long long r_end = r_begin + per * KT;
    if (r_end > g.rows) r_end = g.rows;
    const int nk = r_begin < r_end ? (int)((r_end - r_begin + KT - 1) / KT) : 0;

    // ---- staging maps: Z: thread -> (row tid / 32 + 16 i, 8-column chunk tid % 32);  A (NA = 64): tid < 256 -> (tid / 8, tid % 8)
    const int zq = tid & 31, zr = tid >> 5;
    auto lds_off = [](int r, int q, int ncb) { return ((r >> 2) * ncb + (q >> 1)) * 128 + (r & 3) * 32 + (q & 1) * 16; };
    u32x4 rz[2][2], ra[2][2];
    float cs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cs[i] = 0.f;
    // Loads are UNCONDITIONAL (rows past the slice re-read its last row and are zeroed when staged, a stage past the end re-reads the last stage):
    // with predicated loads or a branch around a fetch the compiler cannot count what is in flight and falls back to `s_waitcnt vmcnt(0)` in front
    // of every stage() - which also waits for the fetch issued just before the MFMAs, i.e. one stage of prefetch instead of two (round 6: the
    // reason k_wgrad<256> sat at 4.98 TB/s, 0.62 of the HBM peak, since round 3).
    const long long r_last = r_end - 1;
    auto fetch = [&](int kt, int set) {
        const long long r0 = r_begin + (long long)(kt < nk ? kt : nk - 1) * KT;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            long long r = r0 + zr + 16 * i;
            r = r < r_last ? r : r_last;
            rz[set][i] = *reinterpret_cast<const u32x4*>(Z + (size_t)r * 512 + zq * 16);
            if constexpr (NA == 256) ra[set][i] = *reinterpret_cast<const u32x4*>(A + (size_t)r * 512 + zq * 16);
        }
        if constexpr (NA == 64) {
            long long r = r0 + (tid >> 3);
            r = r < r_last ? r : r_last;
            if (tid < 256) ra[set][0] = *reinterpret_cast<const u32x4*>(A + (size_t)r * 128 + (tid & 7) * 16);
        }
    };
    auto stage = [&](int kt, int set, int buf) {
        const long long r0 = r_begin + (long long)kt * KT;
        const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool live = r0 + zr + 16 * i < r_end;
            const u32x4 vz = live ? rz[set][i] : zero;
            *reinterpret_cast<u32x4*>(Zs + buf * ZT + lds_off(zr + 16 * i, zq, NCB_Z)) = vz;
            if constexpr (NA == 256) *reinterpret_cast<u32x4*>(As + buf * AT + lds_off(zr + 16 * i, zq, NCB_A)) = live ? ra[set][i] : zero;
            if (r0 + zr + 16 * i < g.cs_rows) {                  // rows past r_end were staged as zeros
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    cs[2 * e] += __uint_as_float(vz[e] << 16);
                    cs[2 * e + 1] += __uint_as_float(vz[e] & 0xffff0000u);
                }
            }
        }
        if constexpr (NA == 64) {
            const bool live = r0 + (tid >> 3) < r_end;
            if (tid < 256) *reinterpret_cast<u32x4*>(As + buf * AT + lds_off(tid >> 3, tid & 7, NCB_A)) = live ? ra[set][0] : zero;
        }
    };

    // ---- this wave's tiles and its lane's read addresses
    const int wm = (NA == 256) ? (w >> 2) : w, wn = (NA == 256) ? (w & 3) : 0;
    const int grp = l >> 4;
    const unsigned lane_z = (unsigned)(((2 * (grp >> 1)) * NCB_Z + (grp & 1) + 2 * MB * wm) * 128 + 8 * (l & 15));
    const unsigned lane_a = (unsigned)(((2 * (grp >> 1)) * NCB_A + (grp & 1) + 2 * NB * wn) * 128 + 8 * (l & 15));
    const unsigned zs0 = (unsigned)(size_t)Zs, as0 = (unsigned)(size_t)As;
    f32x16 acc[MB][NB];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    auto compute = [&](int buf) {
        const unsigned za = zs0 + buf * ZT + lane_z, aa = as0 + buf * AT + lane_a;
        Frags<NA> f0, f1;
        frags_issue<NA, 0>(f0, za, aa);
        frags_wait<NA>(f0);
        frags_issue<NA, 1>(f1, za, aa);          // in flight under k-step 0's MFMAs
        frags_mma<NA>(f0, acc);
        frags_wait<NA>(f1);
        frags_mma<NA>(f1, acc);
    };

    // Two stages per iteration, branch-free between a fetch and the stage() that consumes it (a stage past the end stages zeros into a buffer
    // nobody reads): every path into the loop header carries exactly set 1's four loads, so the compiler's waits are counted (`vmcnt(4)`), never
    // `vmcnt(0)` - each load is in flight across two compute phases.
    if (nk > 0) {
        fetch(0, 0);
        fetch(1, 1);
        stage(0, 0, 0);
        __syncthreads();
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            fetch(kt + 2, 0);
            compute(0);
            stage(kt + 1, 1, 1);
            __syncthreads();
            fetch(kt + 3, 1);
            compute(1);
            stage(kt + 2, 0, 0);
            __syncthreads();
        }
        if (kt < nk) compute(0);                  // odd stage count: the last stage sits in buffer 0
    }
    // ---- partial results: part[mat][sp][p * NA + q], then 256 column sums
    float* out = g.part + ((size_t)mat * g.n_split + sp) * (size_t)(256 * NA + 256);
    // C layout: lane (col n = l & 31, half = l >> 5), reg i -> row m = (i & 3) + 8 (i >> 2) + 4 half
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int p = 32 * (MB * wm + a) + (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
                const int q = 32 * (NB * wn + b) + (l & 31);
                out[(size_t)p * NA + q] = acc[a][b][i];
            }
    // column sums: the 16 threads that staged chunk zq hold partial sums of its 8 columns
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);                // [16][256]
#pragma unroll
    for (int e = 0; e < 8; ++e) red[zr * 256 + zq * 8 + e] = cs[e];
    __syncthreads();
    if (tid < 256) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += red[j * 256 + tid];
        out[(size_t)256 * NA + tid] = s;
    }
}

// out[mat][i] (+)= sum_s part[mat][s][i], i < 256 NA + 256; dW and cs land in their own arrays
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, int n_split, int NA, float* __restrict__ dW,
                                                      float* __restrict__ cs, int accumulate) {
    const int mat = blockIdx.y;
    const int per = 256 * NA + 256;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= per) return;
    const float* p = part + (size_t)mat * n_split * per + i;
    // eight partials in flight per thread (the loop is latency bound otherwise: up to 512 dependent-looking loads, 0.10 ms per call
    // and 19 ms per training step in profiles/r04j_train_kernel_stats.txt); fixed summation order
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 8 <= n_split; k += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(p + (size_t)(k + j) * per);
        s0 += v[0] + v[4]; s1 += v[1] + v[5]; s2 += v[2] + v[6]; s3 += v[3] + v[7];
    }
    for (; k < n_split; ++k) s0 += p[(size_t)k * per];
    const float s = (s0 + s1) + (s2 + s3);
    if (i < 256 * NA) {
        float* o = dW + (size_t)mat * 256 * NA + i;
        *o = accumulate ? *o + s : s;
    } else if (cs) {
        float* o = cs + (size_t)mat * 256 + (i - 256 * NA);
        *o = accumulate ? *o + s : s;
    }
}

static int pick_split(int n_mats, long long rows) {
    // two workgroups per CU in total (one is resident per CU at NA = 256: two rounds, the second hides the first one's tail); every
    // workgroup at least 8 stages
    const long long stages = (rows + KT - 1) / KT;
    long long s = (2LL * num_cus()) / n_mats;
    if (s > stages / 8) s = stages / 8;
    if (s > 512) s = 512;
    return s < 1 ? 1 : (int)s;
}

}  // namespace wgrad
}  // namespace nerfart

using namespace nerfart;
using namespace nerfart::wgrad;

extern "C" {

long long nerfart_wgrad_workspace_bytes(int n_mats, long long rows, int a_cols) {
    if (n_mats <= 0 || rows <= 0 || (a_cols != 256 && a_cols != 64)) return 0;
    return (long long)n_mats * pick_split(n_mats, rows) * (256LL * a_cols + 256) * 4;
}

// dW [n_mats, 256, a_cols] fp32 (row = Z column, unit order as dumped) and cs [n_mats, 256] (NULL: not wanted) from the bf16
// dumps Z [rows, 256] / A [rows, a_cols] of n_mats matrix pairs (z_stride / a_stride bytes apart).  accumulate != 0 adds to the
// outputs (the sum over the patches of a step).  workspace: nerfart_wgrad_workspace_bytes(n_mats, rows, a_cols), caller owned.
int nerfart_wgrad_bf16(const void* Z, long long z_stride, const void* A, long long a_stride, int n_mats, long long rows, int a_cols,
                       long long cs_rows, float* dW, float* cs, int accumulate, void* workspace, long long workspace_bytes, void* stream) {
    if (n_mats <= 0 || rows <= 0) return 0;
    if (a_cols != 256 && a_cols != 64) { set_last_error("wgrad: a_cols must be 256 or 64"); return 2; }
    if (!Z || !A || !dW) { set_last_error("wgrad: null operand"); return 2; }
    const long long need = nerfart_wgrad_workspace_bytes(n_mats, rows, a_cols);
    if (!workspace || workspace_bytes < need) { set_last_error("wgrad: workspace missing or smaller than nerfart_wgrad_workspace_bytes()"); return 2; }
    if (((size_t)Z | (size_t)A | (size_t)z_stride | (size_t)a_stride) & 15) { set_last_error("wgrad: operands must be 16-byte aligned"); return 2; }
    hipStream_t st = (hipStream_t)stream;
    Args g;
    g.Z = (const char*)Z; g.z_stride = z_stride; g.A = (const char*)A; g.a_stride = a_stride;
    g.rows = rows; g.cs_rows = cs ? (cs_rows < rows ? cs_rows : rows) : 0;
    g.n_split = pick_split(n_mats, rows);
    g.part = (float*)workspace;
    const size_t lds = (size_t)2 * KT * 512 + (size_t)2 * KT * a_cols * 2;
    if (a_cols == 256) {
        NERFART_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        // launch profiling class 3 (bench.py's cfg 3 roofline): units = the ALGORITHMIC bytes of the launch, both bf16 operands read once
        void* ph = nullptr;
        if (profile_enabled()) profile_open(3, (long long)n_mats * rows * (256 + 256) * 2, st, &ph);
        hipLaunchKernelGGL(k_wgrad<256>, dim3(g.n_split, n_mats), dim3(THREADS), lds, st, g);
        profile_close(ph, st);
    } else {
        NERFART_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_wgrad<64>, dim3(g.n_split, n_mats), dim3(THREADS), lds, st, g);
    }
    NERFART_HIP(hipGetLastError());
    const int per = 256 * a_cols + 256;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((per + 255) / 256, n_mats), dim3(256), 0, st, (const float*)workspace, g.n_split, a_cols, dW, cs, accumulate);
    NERFART_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
