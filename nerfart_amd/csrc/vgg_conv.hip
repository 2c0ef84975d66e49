// vgg_conv.hip - VGG16 perceptual term of the fine-tune objective (reference criteria/perp_loss.py:9-57): the
// conv stack through relu3_3 on the matrix cores, L1 between the prediction's and the target's features, and the backward pass
// to the prediction's pixels (the VGG weights are frozen).
//
//   conv1_1 (3 -> 64): explicit im2col of the 3-channel input (27 -> 32 columns) + GEMM
//   the other six 3 x 3 convolutions: IMPLICIT GEMM (gemm_f32.h, A_CONV3) on NHWC fp32 activations, bias + ReLU fused;
//   2 x 2 max-pools in between; backward = the same kernel on tap-flipped, channel-transposed weights with the ReLU mask of the
//   layer below fused into the epilogue, un-pooling fused with the mask, col2im gather for conv1_1.
// fp32 operands on v_mfma_f32_32x32x2_f32 (round 3; fp16 operands before): the reference runs torchvision's fp32 VGG, and the
// pixel gradient of this term is piecewise constant in the ReLU / max-pool / sign decisions - half-precision activations flipped
// ~1e-3 of them (gradient 6e-2 off the reference's; now at the level the reference itself moves between thread counts).
#include "gemm_f32.h"
#include <cmath>

namespace nerfart {
namespace vgg {

using namespace nerfart::gemm32;

constexpr int NCONV = 7;
static const int CIN[NCONV] = {3, 64, 64, 128, 128, 256, 256};
static const int COUT[NCONV] = {64, 64, 128, 128, 256, 256, 256};
// resolution level of each conv's input / output: 0 = H x W, 1 = / 2, 2 = / 4 (a 2 x 2 max-pool precedes convs 2 and 4)
static const int LEVEL[NCONV] = {0, 0, 1, 1, 2, 2, 2};

// blob sections (256-byte aligned), all fp32: 3 l + 0: forward weights [Cout, Kf] (l = 0: Kf = 32, column c 9 + ky 3 + kx < 27;
// l > 0: Kf = 9 Cin, column (ky 3 + kx) Cin + c);  3 l + 1: backward weights (l = 0: [64, 64], row k < 27 = W[:, k]^T;
// l > 0: [Cin, 9 Cout], column (ky' 3 + kx') Cout + o = W[o, c, 2 - ky', 2 - kx']);  3 l + 2: bias [Cout].
constexpr int N_SECTIONS = 3 * NCONV;
static long long section_bytes(int i) {
    const int l = i / 3, j = i % 3;
    if (j == 2) return 4LL * COUT[l];
    if (l == 0) return 4LL * 64 * (j == 0 ? 32 : 64);
    return 4LL * 9 * CIN[l] * COUT[l];
}
static long long blob_layout(long long* offs) {
    long long o = 0;
    for (int i = 0; i < N_SECTIONS; ++i) {
        if (offs) offs[i] = o;
        o += (section_bytes(i) + 255) / 256 * 256;
    }
    if (offs) offs[N_SECTIONS] = o;
    return o;
}

// img [B, 3, H, W] fp32 -> cols fp32 [B H W, 32]
__global__ __launch_bounds__(256) void k_im2col_c3(const float* __restrict__ img, float* __restrict__ cols, int B, int H, int W) {
    const long long total = (long long)B * H * W * 32;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int k = (int)(i & 31);
        const long long m = i >> 5;
        float v = 0.f;
        if (k < 27) {
            const int c = k / 9, ky = (k % 9) / 3, kx = k % 3;
            const int x = (int)(m % W), y = (int)((m / W) % H), b = (int)(m / ((long long)W * H));
            const int yy = y + ky - 1, xx = x + kx - 1;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = img[(((size_t)b * 3 + c) * H + yy) * W + xx];
        }
        cols[i] = v;
    }
}
// dcols fp32 [H W, 64] -> g_img [3, H, W] = scale * sum over the 9 windows that contain the pixel
__global__ __launch_bounds__(256) void k_col2im_c3(const float* __restrict__ dcols, float* __restrict__ g_img, int H, int W, const float* __restrict__ scale) {
    const int total = 3 * H * W;
    const float s = scale[0];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int x = i % W, y = (i / W) % H, c = i / (W * H);
        float v = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int oy = y - ky + 1, ox = x - kx + 1;           // the output pixel whose tap (ky, kx) reads (y, x)
                if (oy >= 0 && oy < H && ox >= 0 && ox < W) v += dcols[((size_t)oy * W + ox) * 64 + c * 9 + ky * 3 + kx];
            }
        g_img[i] = v * s;
    }
}
// NHWC fp32 [B, H, W, C] -> [B, H/2, W/2, C]; 4 channels per thread
__global__ __launch_bounds__(256) void k_maxpool2(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
    const long long total = (long long)B * Ho * Wo * C4;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c4 = (int)(i % C4), ox = (int)((i / C4) % Wo), oy = (int)((i / ((long long)C4 * Wo)) % Ho), b = (int)(i / ((long long)C4 * Wo * Ho));
        const float* p = x + (((size_t)b * H + 2 * oy) * W + 2 * ox) * C + c4 * 4;
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), bb = *reinterpret_cast<const f32x4*>(p + C);
        const f32x4 c = *reinterpret_cast<const f32x4*>(p + (size_t)W * C), d = *reinterpret_cast<const f32x4*>(p + (size_t)W * C + C);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = fmaxf(fmaxf(a[k], bb[k]), fmaxf(c[k], d[k]));
        *reinterpret_cast<f32x4*>(y + i * 4) = o;
    }
}
// g [B, H/2, W/2, C] -> dz [B, H, W, C]: to the window's first maximum (row-major scan, as torch), times [y > 0] (the ReLU below)
__global__ __launch_bounds__(256) void k_unpool2_relu(const float* __restrict__ g, const float* __restrict__ yact, float* __restrict__ dz,
                                                     int B, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
    const long long total = (long long)B * Ho * Wo * C4;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c4 = (int)(i % C4), ox = (int)((i / C4) % Wo), oy = (int)((i / ((long long)C4 * Wo)) % Ho), b = (int)(i / ((long long)C4 * Wo * Ho));
        const size_t p0 = (((size_t)b * H + 2 * oy) * W + 2 * ox) * C + c4 * 4;
        const size_t offs[4] = {p0, p0 + C, p0 + (size_t)W * C, p0 + (size_t)W * C + C};
        f32x4 v[4], o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4*>(yact + offs[q]);
        const f32x4 gg = *reinterpret_cast<const f32x4*>(g + i * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int best = 0;
            float m = v[0][k];
#pragma unroll
            for (int q = 1; q < 4; ++q) if (v[q][k] > m) { m = v[q][k]; best = q; }
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q][k] = (q == best && m > 0.f) ? gg[k] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(dz + offs[q]) = o[q];
    }
}
// f [2 n] (prediction's features, then the target's): loss[0] += sum |fp - ft| / n;  g [n] = sign(fp - ft) [fp > 0]
__global__ __launch_bounds__(256) void k_l1_sign(const float* __restrict__ f, long long n, float* __restrict__ loss, float* __restrict__ g) {
    __shared__ float red[4];
    float acc = 0.f;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float a = f[i], b = f[n + i], d = a - b;
        acc += fabsf(d);
        if (g) g[i] = (a > 0.f) ? ((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f)) : 0.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) / (float)n);
}
__global__ void k_set_scale(float* scale, const float* upstream, float inv_n) { scale[0] = (upstream ? upstream[0] : 1.0f) * inv_n; }

static inline long long up(long long v, long long a) { return (v + a - 1) / a * a; }
struct Work {
    int H, W, keep;
    long long M[3];                       // pixels of both images at the three resolutions
    long long o_loss, o_cols, o_y[NCONV], o_p[2], o_ga, o_gb, o_dcols, total;
};
static Work work_layout(int H, int W, int keep) {
    Work w;
    w.H = H; w.W = W; w.keep = keep;
    w.M[0] = 2LL * H * W; w.M[1] = w.M[0] / 4; w.M[2] = w.M[0] / 16;
    long long o = 0;
    auto take = [&](long long b) { const long long at = o; o += up(b, 256); return at; };
    w.o_loss = take(256);
    w.o_cols = take(4 * w.M[0] * 32);
    for (int l = 0; l < NCONV; ++l) w.o_y[l] = take(4 * w.M[LEVEL[l]] * COUT[l]);
    w.o_p[0] = take(4 * w.M[1] * 64);
    w.o_p[1] = take(4 * w.M[2] * 128);
    // backward ping-pong buffers (prediction only: half the rows), sized for the largest cotangent (H W x 64 floats)
    w.o_ga = take(keep ? 4 * (w.M[0] / 2) * 64 : 0);
    w.o_gb = take(keep ? 4 * (w.M[0] / 2) * 64 : 0);
    w.o_dcols = take(keep ? 4 * (w.M[0] / 2) * 64 : 0);
    w.total = o;
    return w;
}
static int check_geo(int H, int W) {
    if (H < 8 || W < 8 || (H & 3) || (W & 3) || ((long long)H * W / 16) % 64) {
        set_last_error("vgg16: H and W must be multiples of 4 with H W / 16 a multiple of 64 (224 x 224 in the reference)");
        return 1;
    }
    return 0;
}
static unsigned grid_for(long long n) { return (unsigned)((n + 255) / 256 < 4096 * 64 ? (n + 255) / 256 : 4096 * 64); }

// ---- packer (ABI 3): torchvision `features.{idx}.weight` [Cout, Cin, 3, 3] / `.bias` -> the three sections of conv l -----------------
__global__ void __launch_bounds__(256) k_pack_conv(const float* __restrict__ w, const float* __restrict__ b, int l, int cin, int cout,
                                                   float* __restrict__ fwd, float* __restrict__ bwd, float* __restrict__ bias) {
    const long long stride = (long long)gridDim.x * blockDim.x, t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (l == 0) {                                   // fwd [64, 32]: column c 9 + ky 3 + kx < 27; bwd [64, 64]: row k < 27 = W[:, k]
        for (long long i = t0; i < 64 * 32; i += stride) { const int o = (int)(i / 32), k = (int)(i % 32); fwd[i] = k < 27 ? w[o * 27 + k] : 0.f; }
        for (long long i = t0; i < 64 * 64; i += stride) { const int k = (int)(i / 64), o = (int)(i % 64); bwd[i] = k < 27 ? w[o * 27 + k] : 0.f; }
    } else {
        const long long n = 9LL * cin * cout;
        for (long long i = t0; i < n; i += stride) {   // fwd [Cout, 9 Cin]: column (ky 3 + kx) Cin + c
            const int o = (int)(i / (9 * cin)), r = (int)(i % (9 * cin)), kk = r / cin, c = r % cin;
            fwd[i] = w[((long long)o * cin + c) * 9 + kk];
        }
        for (long long i = t0; i < n; i += stride) {   // bwd [Cin, 9 Cout]: column (ky' 3 + kx') Cout + o = W[o, c, 2 - ky', 2 - kx']
            const int c = (int)(i / (9 * cout)), r = (int)(i % (9 * cout)), kk = r / cout, o = r % cout;
            bwd[i] = w[((long long)o * cin + c) * 9 + (8 - kk)];
        }
    }
    for (long long i = t0; i < cout; i += stride) bias[i] = b[i];
}

}  // namespace vgg
}  // namespace nerfart

using namespace nerfart;
using namespace nerfart::vgg;

extern "C" {

long long nerfart_vgg16_blob_layout(long long* offsets) { return blob_layout(offsets); }

// The packer of that blob: weight[l] [Cout_l, Cin_l, 3, 3], bias[l] [Cout_l] (HOST arrays of 7 DEVICE fp32 pointers: torchvision vgg16
// `features.{0, 2, 5, 7, 10, 12, 14}`, the convolutions up to relu3_3, criteria/perp_loss.py:9-33) -> forward / backward / bias sections.
int nerfart_vgg16_pack(const float* const* weight, const float* const* bias, void* blob, long long blob_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!weight || !bias || !blob) { set_last_error("vgg16_pack: NULL argument"); return 2; }
    if (blob_bytes != blob_layout(nullptr)) { set_last_error("vgg16_pack: blob_bytes differs from nerfart_vgg16_blob_layout()"); return 1; }
    long long off[N_SECTIONS + 1];
    blob_layout(off);
    NERFART_HIP(hipMemsetAsync(blob, 0, (size_t)blob_bytes, st));
    char* bl = (char*)blob;
    for (int l = 0; l < NCONV; ++l) {
        if (!weight[l] || !bias[l]) { set_last_error("vgg16_pack: NULL tensor pointer"); return 2; }
        hipLaunchKernelGGL(k_pack_conv, dim3(1024), dim3(256), 0, st, weight[l], bias[l], l, CIN[l], COUT[l], (float*)(bl + off[3 * l]),
                           (float*)(bl + off[3 * l + 1]), (float*)(bl + off[3 * l + 2]));
    }
    NERFART_HIP(hipGetLastError());
    return 0;
}
long long nerfart_vgg16_workspace_bytes(int H, int W, int keep_for_bwd) { return (H >= 8 && W >= 8) ? work_layout(H, W, keep_for_bwd).total : 0; }

// img2 [2, 3, H, W] fp32: the normalised (and resized) prediction, then the target.  loss_out[0] = mean |relu3_3(pred) - relu3_3(target)|.
int nerfart_vgg16_l1_fwd(const void* blob, long long blob_bytes, const float* img2, int H, int W, float* loss_out, int keep_for_bwd, void* workspace,
                         long long workspace_bytes, void* stream) {
    if (check_geo(H, W)) return 1;
    if (blob && blob_bytes != blob_layout(nullptr)) { set_last_error("vgg16_l1_fwd: blob_bytes differs from nerfart_vgg16_blob_layout(): blob packed to another layout / ABI version"); return 1; }
    const Work w = work_layout(H, W, keep_for_bwd);
    if (!blob || !workspace || workspace_bytes < w.total) { set_last_error("vgg16_l1_fwd: null blob / workspace too small"); return 1; }
    hipStream_t st = (hipStream_t)stream;
    long long off[N_SECTIONS + 1];
    blob_layout(off);
    const char* bl = (const char*)blob;
    char* ws = (char*)workspace;
    float* loss = (float*)(ws + w.o_loss);
    NERFART_HIP(hipMemsetAsync(loss, 0, 256, st));
    float* cols = (float*)(ws + w.o_cols);
    hipLaunchKernelGGL(k_im2col_c3, dim3(grid_for(w.M[0] * 32)), dim3(256), 0, st, img2, cols, 2, H, W);
    const float* x = cols;
    int h = H, wd = W;
    for (int l = 0; l < NCONV; ++l) {
        float* y = (float*)(ws + w.o_y[l]);
        if (l == 2 || l == 4) {                                    // max-pool in front of conv2_1 / conv3_1
            float* p = (float*)(ws + w.o_p[l == 2 ? 0 : 1]);
            hipLaunchKernelGGL(k_maxpool2, dim3(grid_for(w.M[LEVEL[l]] * CIN[l] / 4)), dim3(256), 0, st, x, p, 2, h, wd, CIN[l]);
            x = p; h /= 2; wd /= 2;
        }
        Epi e{};
        e.bias = (const float*)(bl + off[3 * l + 2]); e.out = y; e.ldo = COUT[l]; e.m_valid = (int)w.M[LEVEL[l]];
        e.cH = h; e.cW = wd; e.cC = CIN[l];
        const float* Wf = (const float*)(bl + off[3 * l]);
        const int rc = (l == 0) ? gemm<EPI_BIAS_RELU, A_MAT>(st, x, 32, Wf, (int)w.M[0], 64, 32, e)
                                : gemm<EPI_BIAS_RELU, A_CONV3>(st, x, 0, Wf, (int)w.M[LEVEL[l]], COUT[l], 9 * CIN[l], e);
        if (rc) return 1;
        x = y;
    }
    const long long n = (w.M[2] / 2) * 256;
    hipLaunchKernelGGL(k_l1_sign, dim3(grid_for(n)), dim3(256), 0, st, x, n, loss, keep_for_bwd ? (float*)(ws + w.o_ga) : (float*)nullptr);
    NERFART_HIP(hipMemcpyAsync(loss_out, loss, sizeof(float), hipMemcpyDeviceToDevice, st));
    NERFART_HIP(hipGetLastError());
    return 0;
}

// g_img [1, 3, H, W] = upstream[0] (device scalar; NULL = 1) * d loss / d (normalised prediction), from the kept workspace.
int nerfart_vgg16_l1_bwd(const void* blob, long long blob_bytes, int H, int W, const float* upstream, float* g_img, void* workspace, long long workspace_bytes, void* stream) {
    if (check_geo(H, W)) return 1;
    if (blob && blob_bytes != blob_layout(nullptr)) { set_last_error("vgg16_l1_bwd: blob_bytes differs from nerfart_vgg16_blob_layout(): blob packed to another layout / ABI version"); return 1; }
    const Work w = work_layout(H, W, 1);
    if (!blob || !workspace || workspace_bytes < w.total) { set_last_error("vgg16_l1_bwd: null blob / workspace too small"); return 1; }
    hipStream_t st = (hipStream_t)stream;
    long long off[N_SECTIONS + 1];
    blob_layout(off);
    const char* bl = (const char*)blob;
    char* ws = (char*)workspace;
    float* scale = (float*)(ws + w.o_loss) + 8;
    hipLaunchKernelGGL(k_set_scale, dim3(1), dim3(1), 0, st, scale, upstream, 1.0f / (float)((w.M[2] / 2) * 256));
    float* ga = (float*)(ws + w.o_ga);                             // holds sign * mask of relu3_3 (written by the forward)
    float* gb = (float*)(ws + w.o_gb);
    int h = H / 4, wd = W / 4;
    for (int l = NCONV - 1; l >= 1; --l) {
        // cotangent of conv l's output (already masked by its ReLU) -> cotangent of its input
        const long long Mi = w.M[LEVEL[l]] / 2;
        Epi e{};
        e.out = gb; e.ldo = CIN[l]; e.m_valid = (int)Mi; e.cH = h; e.cW = wd; e.cC = COUT[l];
        const float* Wb = (const float*)(bl + off[3 * l + 1]);
        const bool pooled_input = (l == 2 || l == 4);
        int rc;
        if (pooled_input) {
            rc = gemm<EPI_PLAIN, A_CONV3>(st, ga, 0, Wb, (int)Mi, CIN[l], 9 * COUT[l], e);
        } else {
            e.aux = (const float*)(ws + w.o_y[l - 1]);             // prediction rows come first in every activation buffer
            rc = gemm<EPI_RELUMASK, A_CONV3>(st, ga, 0, Wb, (int)Mi, CIN[l], 9 * COUT[l], e);
        }
        if (rc) return 1;
        if (pooled_input) {                                        // un-pool into the activation below and apply its ReLU mask
            h *= 2; wd *= 2;
            hipLaunchKernelGGL(k_unpool2_relu, dim3(grid_for(Mi * CIN[l] / 4)), dim3(256), 0, st, gb, (const float*)(ws + w.o_y[l - 1]), ga, 1, h, wd, CIN[l]);
        } else {
            float* t = ga; ga = gb; gb = t;
        }
    }
    // conv1_1: d cols = g1 W0 (27 real columns), then the col2im gather
    float* dcols = (float*)(ws + w.o_dcols);
    { Epi e{}; e.out = dcols; e.ldo = 64; e.m_valid = (int)(w.M[0] / 2);
      if (gemm<EPI_PLAIN, A_MAT>(st, ga, 64, (const float*)(bl + off[1]), (int)(w.M[0] / 2), 64, 64, e)) return 1; }
    hipLaunchKernelGGL(k_col2im_c3, dim3(grid_for(3LL * H * W)), dim3(256), 0, st, dcols, g_img, H, W, scale);
    NERFART_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
