// mlp_common.h - pieces shared by the fp32 (mlp_chain.hip) and split-bf16 (mlp_chain_bf16.hip) chained-MLP
// kernels: the point source (explicit points or rays + depths) and the L2 -> LDS weight-chunk pipeline.
#pragma once
#include "nerfart_common.h"

namespace nerfart {

constexpr int TAB_INTS = 128;                       // chunk offset table (nc + 1 <= 128 entries)

// Where the points of a launch come from: an explicit [M,3] array, or rays + per-ray depths
// (point m = slot m / n_per_ray, sample m % n_per_ray; ray = ray_idx ? ray_idx[slot] : slot).
struct PointSrc {
    const float* pts;
    const float* rays_o;
    const float* rays_d;
    const int* ray_idx;
    const float* depth;
    const float* view;      // explicit per-point view dirs [M,3] (pts mode, radiance only)
    int n_per_ray;
    int depth_stride;
    unsigned M;
};

struct Pt { float x, y, z, vx, vy, vz; };

__device__ __forceinline__ Pt fetch_point(const PointSrc& s, unsigned m, bool want_view) {
    Pt p = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (m >= s.M) return p;
    if (s.pts) {
        p.x = s.pts[3 * (size_t)m + 0]; p.y = s.pts[3 * (size_t)m + 1]; p.z = s.pts[3 * (size_t)m + 2];
        if (want_view) { p.vx = s.view[3 * (size_t)m + 0]; p.vy = s.view[3 * (size_t)m + 1]; p.vz = s.view[3 * (size_t)m + 2]; }
    } else {
        const unsigned slot = m / (unsigned)s.n_per_ray;
        const unsigned k = m - slot * (unsigned)s.n_per_ray;
        const unsigned ray = s.ray_idx ? (unsigned)s.ray_idx[slot] : slot;
        const float t = s.depth[(size_t)slot * s.depth_stride + k];
        const float ox = s.rays_o[3 * (size_t)ray + 0], oy = s.rays_o[3 * (size_t)ray + 1], oz = s.rays_o[3 * (size_t)ray + 2];
        p.vx = s.rays_d[3 * (size_t)ray + 0]; p.vy = s.rays_d[3 * (size_t)ray + 1]; p.vz = s.rays_d[3 * (size_t)ray + 2];
        p.x = ray_point(ox, p.vx, t); p.y = ray_point(oy, p.vy, t); p.z = ray_point(oz, p.vz, t);
    }
    return p;
}

// ---------------------------------------------------------------------------------------
// Weight-chunk pipeline: chunk c of the blob is consumed from LDS buffer pb while chunk c+1
// streams into buffer pb^1.  The chunk offset table lives in LDS (copied once at kernel start);
// the offsets of the chunk to prefetch are looked up one acquire ahead so the lookup latency never
// sits between the barrier and the LDS-DMA issue.  (Reading the table from global memory here
// would be a VMEM load - the LDS-DMA asm makes hipcc treat global memory as clobbered, so it
// cannot use scalar loads - and its vmcnt(0) wait would serialise behind the DMA.)
// ---------------------------------------------------------------------------------------
template <int NWAVES, int CHUNK_FLOATS>
struct PipeT {
    const float* blob;
    const int* tab;   // LDS copy of header[NERFART_HDR_OFFS ...]: nc + 1 float offsets
    float* lds;
    int nc;           // chunks per pass
    int nxt;          // chunk the next acquire() will prefetch (-1: none)
    int nxt_o0, nxt_o1;
    int pb;           // LDS buffer the next acquire() returns
    bool wrap;        // another tile follows: prefetch chunk 0 after the last chunk
};

template <int NWAVES, int CHUNK_FLOATS>
__device__ __forceinline__ void pipe_issue_range(const PipeT<NWAVES, CHUNK_FLOATS>& p, int o0, int o1, int buf) {
    const int npieces = (o1 - o0) >> 8;                       // 1 KiB (256 floats) per wave-instruction
    const float* src = p.blob + o0 + lane_id() * 4;
    const unsigned dst = lds_addr(p.lds + buf * CHUNK_FLOATS);
#ifndef NERFART_ABLATE_DMA      // timing experiments only (tools/ablate_bf16.sh): skip the weight stream
    for (int q = wave_id(); q < npieces; q += NWAVES)
        glds16(src + q * 256, __builtin_amdgcn_readfirstlane(dst + q * 1024));
#else
    (void)src; (void)dst; (void)npieces;
#endif
}

template <int NWAVES, int CHUNK_FLOATS>
__device__ __forceinline__ void pipe_lookup(PipeT<NWAVES, CHUNK_FLOATS>& p, int chunk) {
    p.nxt = chunk;
    if (chunk >= 0) {
        p.nxt_o0 = __builtin_amdgcn_readfirstlane(p.tab[chunk]);
        p.nxt_o1 = __builtin_amdgcn_readfirstlane(p.tab[chunk + 1]);
    }
}

// Called once per workgroup after the table is in LDS: start streaming chunk 0 into buffer 0.
template <int NWAVES, int CHUNK_FLOATS>
__device__ __forceinline__ void pipe_start(PipeT<NWAVES, CHUNK_FLOATS>& p) {
    pipe_lookup(p, 0);
    pipe_issue_range(p, p.nxt_o0, p.nxt_o1, 0);
    p.pb = 0;
    pipe_lookup(p, 1 < p.nc ? 1 : -1);
}

template <int NWAVES, int CHUNK_FLOATS>
__device__ __forceinline__ const float* pipe_acquire(PipeT<NWAVES, CHUNK_FLOATS>& p) {
    wait_glds();          // my pieces of the current chunk have landed
    __syncthreads();      // everyone's pieces landed; everyone is done with buffer pb^1
    const int cur = p.nxt;
    if (cur >= 0) pipe_issue_range(p, p.nxt_o0, p.nxt_o1, p.pb ^ 1);
    int f = cur + 1;
    if (cur < 0) f = -1;
    else if (f == p.nc) f = p.wrap ? 0 : -1;
    pipe_lookup(p, f);
    const float* w = p.lds + p.pb * CHUNK_FLOATS;
    p.pb ^= 1;
    return w;
}


// ---- host-side helpers -------------------------------------------------------------------
inline int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return n;
}

inline int validate_src(const PointSrc& s) {
    if (s.M == 0) return 0;
    if (!s.pts && !(s.rays_o && s.rays_d && s.depth && s.n_per_ray > 0)) {
        set_last_error("point source: need pts, or rays_o + rays_d + depth + n_per_ray");
        return 2;
    }
    return 0;
}

inline PointSrc make_src(const float* pts, const float* view, const float* rays_o, const float* rays_d,
                         const int* ray_idx, const float* depth, int n_per_ray, int depth_stride, long long M) {
    PointSrc s;
    s.pts = pts; s.view = view; s.rays_o = rays_o; s.rays_d = rays_d; s.ray_idx = ray_idx; s.depth = depth;
    s.n_per_ray = n_per_ray; s.depth_stride = depth_stride; s.M = (unsigned)M;
    return s;
}

inline int check_M(long long M) {
    if (M < 0 || M >= (1ll << 31)) { set_last_error("M out of range (0 <= M < 2^31 points per launch)"); return 2; }
    return 0;
}

}  // namespace nerfart
