// raygen.hip - pixel grid -> world rays (reference utils/rend_util.py: lift :95-109, get_rays :112-165).
#include "nerfart_common.h"

namespace nerfart {

// One thread per selected pixel.  Pixel (col i, row j) carries NO half-pixel offset; ray index
// = j * W + i (rend_util.py:126-128).  pose: row-major 4x4 camera-to-world; K: row-major 4x4.
__global__ void k_get_rays(const float* __restrict__ pose, const float* __restrict__ K, int H, int W,
                           const long long* __restrict__ select, int n, float* __restrict__ rays_o,
                           float* __restrict__ rays_d) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const long long pix = select ? select[idx] : idx;
    const float i = (float)(pix % W), j = (float)(pix / W);
    const float fx = K[0], sk = K[1], cx = K[2], fy = K[5], cy = K[6];
    // x = (i - cx + cy*sk/fy - sk*j/fy) / fx * z ;  y = (j - cy) / fy * z ;  z = 1   (rend_util.py:105-106)
    const float x = ((i - cx + cy * sk / fy) - sk * j / fy) / fx;
    const float y = (j - cy) / fy;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float wr = pose[4 * r] * x + pose[4 * r + 1] * y + pose[4 * r + 2] + pose[4 * r + 3];
        rays_d[3 * (size_t)idx + r] = wr - pose[4 * r + 3];          // world - cam_loc (rend_util.py:160)
        rays_o[3 * (size_t)idx + r] = pose[4 * r + 3];
    }
}

}  // namespace nerfart

extern "C" int nerfart_get_rays(const float* pose_dev, const float* K_dev, int H, int W, const long long* select_dev,
                                int n, float* rays_o, float* rays_d, void* stream) {
    if (n <= 0) return 0;
    if (H <= 0 || W <= 0) { nerfart::set_last_error("get_rays: bad image size"); return 2; }
    hipLaunchKernelGGL(nerfart::k_get_rays, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pose_dev, K_dev, H, W,
                       select_dev, n, rays_o, rays_d);
    NERFART_HIP(hipGetLastError());
    return 0;
}
