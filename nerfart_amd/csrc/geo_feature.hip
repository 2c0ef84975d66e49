// geo_feature.hip - the geometry feature of ImplicitSurface.forward(x, return_h=True) / forward_with_nablas (models/base.py:243-282): rows 1..256 of the
// SDF net's last linear layer applied to the layer-7 activations,
//     feat[m, :] = W8[1:, :] h7[m, :] + b8[1:],      W8 = weight_g * weight_v / ||weight_v||  (nn.utils.weight_norm, dim 0).
// The frame path never materialises it (the radiance kernels evaluate these rows from their own blob, mlp_bf16_core.h); this entry point serves the
// reference's OTHER consumers of the callable (boundary B3: mesh / feature probes) without a library GEMM: fold (ATen's summation order) + the fp32
// MFMA GEMM of gemm_f32.h (v_mfma_f32_32x32x2_f32: exact fp32 products - the reference computes this layer in fp32).
#include "gemm_f32.h"

namespace nerfart {

// W[o][k] = (g[o + 1] * v[o + 1][k]) * (1 / ||v[o + 1]||), o < 256: one block per row, the norm as aten's weight_norm_fwd_first_dim_kernel sums it
__global__ void __launch_bounds__(256) k_fold_feature_rows(const float* __restrict__ g, const float* __restrict__ v, float* __restrict__ W) {
    __shared__ float x[256];
    const int tid = threadIdx.x, row = blockIdx.x + 1;
    const float val = v[(size_t)row * 256 + tid];
    x[tid] = val * val;
    __syncthreads();
    if (tid < 128) x[tid] = x[tid] + x[tid + 128];
    __syncthreads();
    if (tid < 64) x[tid] = x[tid] + x[tid + 64];
    __syncthreads();
    if (tid < 32) {
        float fin = x[tid] + x[tid + 32];
        for (int i = 16; i >= 1; i >>= 1) fin = fin + __shfl_down(fin, i);
        if (tid == 0) x[0] = 1.f / sqrtf(fin);
    }
    __syncthreads();
    W[(size_t)blockIdx.x * 256 + tid] = (g[row] * val) * x[0];
}

}  // namespace nerfart

using namespace nerfart;

extern "C" {

long long nerfart_geometry_feature_workspace_bytes(void) { return 256LL * 256 * 4; }

// weight_g [257] (or [257, 1]), weight_v [257, 256], bias [257]: `implicit_surface.surface_fc_layers.8.*`; h7 [M, 256] fp32 (nerfart_sdf_nabla_fwd's
// h7 output); feat_out [M, 256].  workspace: nerfart_geometry_feature_workspace_bytes() (the folded rows).
int nerfart_geometry_feature(const float* weight_g, const float* weight_v, const float* bias, const float* h7, long long M, float* feat_out,
                             void* workspace, long long workspace_bytes, void* stream) {
    if (M <= 0) return 0;
    if (!weight_g || !weight_v || !bias || !h7 || !feat_out) { set_last_error("geometry_feature: null pointer"); return 2; }
    if (!workspace || workspace_bytes < nerfart_geometry_feature_workspace_bytes()) { set_last_error("geometry_feature: workspace smaller than nerfart_geometry_feature_workspace_bytes()"); return 2; }
    if (M > 2147483647LL - 64) { set_last_error("geometry_feature: M too large for one launch"); return 2; }
    hipStream_t st = (hipStream_t)stream;
    float* W = (float*)workspace;
    hipLaunchKernelGGL(k_fold_feature_rows, dim3(256), dim3(256), 0, st, weight_g, weight_v, W);
    NERFART_HIP(hipGetLastError());
    gemm32::Epi e{};
    e.bias = bias + 1; e.out = feat_out; e.aux = nullptr; e.ldo = 256; e.m_valid = (int)M;
    const int Mp = (int)((M + 63) / 64 * 64);
    return gemm32::gemm<gemm32::EPI_BIAS, gemm32::A_MAT>(st, h7, 256, W, Mp, 256, 256, e);
}

}  // extern "C"
