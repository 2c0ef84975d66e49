/* nerfart_hip.h - C ABI of libnerfart_hip.so: the MI355X (gfx950) hot path of cassiePython/NeRF-Art.
 *
 * The reference (pure Python/PyTorch, no FFI of its own) exposes this path as Python callables;
 * each entry point below names the reference callable (file:line under the reference tree) it
 * replaces.  SURVEY.md section 8b lists the boundaries B1-B3; INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 (int32 / int64 where typed so) unless a
 *     parameter is documented as host; outputs are caller-allocated; nothing is retained.
 *   - `stream` is a hipStream_t (0 = default stream); all work is enqueued on it.  Entry points
 *     with a data-dependent loop (fine_sample, render) synchronise that stream internally.
 *   - return value 0 = success; non-zero = failure, message via nerfart_last_error() (thread local).
 *   - `blob` arguments are weight blobs produced by nerfart_amd/packing.py (layout documented
 *     there and in DESIGN.md): `surf_blob` = SDF net, `rad_blob` = geometry-feature rows + radiance net.
 *   - `precision` selects the matrix-core path AND the blob format it expects:
 *       0 = fp32-exact (v_mfma_f32_16x16x4_f32; blobs from surface_plan()/radiance_plan()).  nerfart_sdf_nabla_fwd* runs
 *           REVERSE mode (k_sdf_grad: forward sweep + transposed chunks of the same blob, softplus' parked as fp32 in the
 *           caller's workspace); precision 3 (these two entry points only) selects the forward-mode tangent quads of
 *           k_sdf_nabla on the same blob (cross-checks),
 *       1 = split bf16 "bf16x3" (3 x v_mfma_f32_16x16x32_bf16 per k-step on hi/lo operand splits, ~2^-16
 *           relative per product; blobs from surface_plan_bf16()/radiance_plan_bf16()).  nerfart_sdf_nabla_fwd*
 *           then runs REVERSE mode too (forward sweep + transposed-weight sweep in one kernel, softplus' as unorm16 in the
 *           caller's workspace); precision 2 (these two entry points only) selects the forward-mode tangent kernel on the
 *           same blob (cross-checks).
 *       4 = "fp16x2" (csrc/mlp_chain_f16x2.hip): the split-bf16 kernels' data flow with the 2-MFMA split - ONE fp16 activation term
 *           x fp16 hi + lo weight terms, 2 x v_mfma_f32_16x16x32_f16 per product (11-bit activations, TF32 class; blobs from
 *           surface_plan_bf16(term="fp16") / radiance_plan_bf16(term="fp16")).  A MEASUREMENT variant of the three forward
 *           kernels (inference entry points only; workspace as precision 1), never a default - DESIGN.md 4.1b.  It is the arithmetic of
 *           Algorithm 1's sampler in the shipped `mixed` mode (nerfart_volsdf_render_staged_fwd's sampler_precision, guarded).
 *       5 = "fp16x1" (csrc/mlp_chain_f16x1.hip; nerfart_sdf_fwd / nerfart_sdf_fwd_rays and the SAMPLER arguments of nerfart_volsdf_fine_sample[_guarded] /
 *           nerfart_volsdf_render_mixed_fwd / _staged_fwd only - every other entry point refuses it): K2 with ONE v_mfma_f32_16x16x32_f16 per
 *           product (one fp16 activation term x one fp16 weight term) in a SCALED softplus recursion (accumulators z' = c z, activations a' = c softplus(z),
 *           c = 100 log2 e).  Its blob: nerfart_pack_surface_blob(precision = 5, ...) - the precision-4 layout under encoding word 3 - from tensors the CALLER
 *           hands over in that recursion (layer 0's weights, the skip layer's encoding columns and the hidden biases times c, the last layer's rows / c) with the
 *           hidden layers' folded weights on the fp16 grid; for rendering, ERROR-COMPENSATED ones (nerfart_amd/calibrate.py).  The library
 *           refuses a precision-4 blob at precision 5 and vice versa.  No value that reaches a pixel and no gradient is computed in it.
 *   - point sources: either an explicit array pts[M,3], or ("_rays" variants) rays + per-ray depths:
 *     point m = slot m / n_per_ray, sample m % n_per_ray, ray = ray_idx ? ray_idx[slot] : slot,
 *     x = rays_o[ray] + rays_d[ray] * depth[slot * depth_stride + sample]  (two roundings, as the
 *     reference's separate mul and add).
 */
#ifndef NERFART_HIP_H
#define NERFART_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* 5 (round 6, second session).  History: 4 -> 5, additions only: nerfart_volsdf_fine_sample_guarded2 / nerfart_volsdf_render_staged2_fwd (the guarded sampler's
 * `late_round`), C-ABI precision 5 on the SDF-only entry points.
 * 3 -> 4, additions only: the guarded sampler (nerfart_volsdf_fine_sample_guarded) and the per-stage-precision renderer
 * (nerfart_volsdf_render_staged_fwd), nerfart_pack_layer_dims, nerfart_geometry_feature.
 * 2 -> 3, additions only (every version-2 entry point keeps its signature): the weight-blob packers
 * (nerfart_pack_surface_blob / nerfart_pack_radiance_blob + size queries; header word 10 of a split blob now names its fragment encoding),
 * nerfart_neus_render_algo_fwd / nerfart_neus_direct_upsample_step (upsample_algo 'direct_use' / 'direct_more').  1 -> 2: nerfart_sdf_nabla_fwd[_rays] take a caller-owned workspace; the CLIP blob stores every matrix once; the
 * VGG blob is fp32; the CLIP / VGG entry points take blob_bytes and reject blobs of another layout; new: the ray-level backward
 * (nerfart_*_render_bwd, nerfart_sdf_param_bwd, nerfart_fold_weight_grads, nerfart_weight_norm_bwd). */
int nerfart_abi_version(void);
const char* nerfart_last_error(void);

/* Optional launch profiling (bench.py): between begin and end every chained-MLP launch and every 256-column weight-gradient
 * launch is bracketed by HIP events on its own stream; end() returns per kernel class c = 0 k_sdf_only, 1 k_sdf_nabla / k_sdf_grad,
 * 2 k_radiance (units = points processed), 3 k_wgrad<256> (units = algorithmic bytes: both bf16 operands read once) the summed
 * elapsed ms, launch count and units (host arrays of 4). */
int nerfart_profile_begin(void);
int nerfart_profile_end(double* ms, long long* launches, long long* units);
/* (ABI 4) The same with host arrays of 5: class 4 = the SDF queries of the guarded sampler's ESCALATION run (nerfart_volsdf_fine_sample_guarded: the rays
 * sampled again, on the escalation blob) - kept out of class 0, so that class 0 is the dominant kernel's own launches; nerfart_profile_end drops it. */
int nerfart_profile_end5(double* ms, long long* launches, long long* units);

/* host helper: torch.linspace(start, end, n) in fp32, bit for bit (out is a HOST array). */
void nerfart_linspace(float start, float end, int n, float* out);

/* ---- B3: SDF query.  ImplicitSurface.forward (models/base.py:243-263) + VolSDF.forward_surface
 * sphere clamp sdf = min(sdf, R_bg - |x|) (models/frameworks/volsdf.py:341-347); R_bg <= 0: no clamp
 * (NeuS, neus.py:277,299). */
int nerfart_sdf_fwd(const float* surf_blob, int precision, const float* pts, long long M, float R_bg, float* sdf_out, void* stream);
int nerfart_sdf_fwd_rays(const float* surf_blob, int precision, const float* rays_o, const float* rays_d, const int* ray_idx,
                         const float* depth, int n_slots, int n_per_ray, int depth_stride, float R_bg,
                         float* sdf_out, int out_stride, void* stream);

/* ---- B2 (first half): ImplicitSurface.forward_with_nablas (models/base.py:265-282) + the clamp of
 * VolSDF.forward_surface_with_nablas (volsdf.py:349-357; nabla is NOT replaced).  Outputs sdf[M],
 * nabla[M,3] and, if h7_out != NULL, the layer-7 activation h7[M,256] consumed by nerfart_radiance_fwd.
 * workspace: nerfart_sdf_nabla_workspace_bytes(precision) bytes of device memory owned by the CALLER (PyTorch's caching
 * allocator in the Python host), uninitialised, private to the call until the stream has passed it: the
 * reverse-mode kernels park softplus'(z_l) of the forward sweep there (one region per resident workgroup; precisions 2 / 3
 * need none and accept NULL).  The library allocates no device memory. */
long long nerfart_sdf_nabla_workspace_bytes(int precision);
int nerfart_sdf_nabla_fwd(const float* surf_blob, int precision, const float* pts, long long M, float R_bg, float* sdf_out,
                          float* nabla_out, float* h7_out, void* workspace, long long workspace_bytes, void* stream);
int nerfart_sdf_nabla_fwd_rays(const float* surf_blob, int precision, const float* rays_o, const float* rays_d, const int* ray_idx,
                               const float* depth, int n_slots, int n_per_ray, int depth_stride, float R_bg,
                               float* sdf_out, float* nabla_out, float* h7_out, void* workspace, long long workspace_bytes, void* stream);

/* ---- B2 (second half): geometry feature (last SDF layer rows 1..256, base.py:253-256) + RadianceNet.forward
 * (models/base.py:372-391) on [x, view (raw: view_tiles=1 | embed 4: view_tiles=3), nabla, feat]. */
int nerfart_radiance_fwd(const float* rad_blob, int precision, int view_tiles, const float* pts, const float* view, long long M,
                         const float* nabla, const float* h7, float* rgb_out, void* stream);
int nerfart_radiance_fwd_rays(const float* rad_blob, int precision, int view_tiles, const float* rays_o, const float* rays_d,
                              const int* ray_idx, const float* depth, int n_slots, int n_per_ray, int depth_stride,
                              const float* nabla, const float* h7, float* rgb_out, void* stream);

/* ---- B2 "bwd", radiance half (row a19; split-bf16 blobs only).  nerfart_radiance_fwd_dump = nerfart_radiance_fwd
 * that also writes the activations of the five layers (geometry feature f, four ReLU outputs r0..r3) into `dump`
 * (nerfart_radiance_dump_bytes(M) bytes): a bf16 matrix [5][Mp][256], Mp = M rounded up to 128, rows = points, the 256
 * features in the kernels' unit order (nerfart_amd/packing.py: unit_feature_hidden) - GEMM operands, read in place.
 * nerfart_radiance_bwd: d loss / d rgb[M,3] -> g_h7[M,256] (cotangent of the SDF net's layer-7 activation through the
 * geometry-feature rows), g_n[M,3] (cotangent of the normal input), and bwd_dump (same layout: the deltas of
 * R0, R1, R2, R3 - each in the slot of the activation it multiplies - and the geometry-feature cotangent): the operands
 * of the weight-gradient reductions (dW_l = delta_l^T act_{l-1}: nerfart_wgrad_bf16 below; call sequence in nerfart_amd/autodiff.py).
 * At most 2^21 points per call. */
long long nerfart_radiance_dump_bytes(long long M);
int nerfart_radiance_fwd_dump(const float* rad_blob, int view_tiles, const float* pts, const float* view, long long M,
                              const float* nabla, const float* h7, float* rgb_out, void* dump, void* stream);
int nerfart_radiance_bwd(const float* rad_blob, long long M, const float* rgb, const float* g_rgb, void* fwd_dump, void* bwd_dump,
                         float* g_h7_out, float* g_n_out, void* stream);

/* ---- B2 "bwd", SDF half (row a19; split-bf16 surface blob).  Parameter gradients of
 *        sbar * sdf + hbar7 . h7 + nbar . grad_x sdf        (what rgb.backward + eikonal.backward ask of the SDF net,
 * volsdf.py:766-770, through ImplicitSurface.forward_with_nablas' create_graph=True, base.py:272-279) need, per layer,
 * the operands of  dW_l = sum_p zbar_l (x) a_{l-1} + (t_l d_l) (x) adot_{l-1}  (see csrc/mlp_chain_bf16.hip):
 *   nerfart_sdf_fwd2: pts[M,3], dir[M,3] = nbar -> f2_dump: [a_l; adot_l] bf16 (slots l = 0..7) and softplus'(z_l) unorm16
 *                     (slots 8 + l)
 *   nerfart_sdf_bwd2: gbar_h7[M,256], gbar_sdf[M], f2_dump -> r2_dump: 65535 * [zbar_l; t_l d_l] bf16, l = 0..7
 * Both dumps are matrices [slot][2 Mp][256], Mp = M rounded up to 64: rows 0..Mp-1 of a slot hold the first, rows Mp..
 * the second quantity of the pair, per point; features in unit order; so dW_l is ONE GEMM over the stacked rows, read in
 * place.  (The softplus' slots 8..15 of f2_dump hold one value per point: rows 0..Mp-1 only, rows Mp.. are never written or read.)
 * At most 2^21 points per call.
 * The reductions are nerfart_wgrad_bf16 calls; the whole sequence - these two sweeps, the reductions, the un-permutation and the weight_norm chain
 * rule - is behind nerfart_sdf_param_bwd / nerfart_*_render_bwd + nerfart_fold_weight_grads / nerfart_weight_norm_bwd (csrc/render_backward.hip). */
long long nerfart_sdf_fwd2_dump_bytes(long long M);
long long nerfart_sdf_bwd2_dump_bytes(long long M);
int nerfart_sdf_fwd2(const float* surf_blob, const float* pts, const float* dir, long long M, void* f2_dump, void* stream);
int nerfart_sdf_bwd2(const float* surf_blob, long long M, const float* gbar_h7, const float* gbar_sdf, void* f2_dump, void* r2_dump,
                     void* stream);

/* ---- rays.  rend_util.get_rays (utils/rend_util.py:112-165) for one camera: pose/K are row-major 4x4
 * on the device; select (int64, may be NULL = all H*W pixels in row-major order) picks pixel indices. */
int nerfart_get_rays(const float* pose_dev, const float* K_dev, int H, int W, const long long* select_dev, int n,
                     float* rays_o, float* rays_d, void* stream);
/* F.normalize(rays_d, dim=-1) (volsdf.py:442, neus.py:196) */
int nerfart_normalize_dirs(const float* in, float* out, int n, void* stream);
/* out[r, k] = near[r]*(1 - t[k]) + far[r]*t[k] (volsdf.py:472-484, neus.py:235-236); near/far NULL -> scalars */
int nerfart_linspace_depths(const float* t_dev, int n, const float* near, const float* far, float near_s, float far_s,
                            int n_rays, float* out, int stride, void* stream);

/* ---- VolSDF sampler stages (models/frameworks/volsdf.py fine_sample :97-302; error_bound :56-94;
 * utils/rend_util.py sample_pdf :256-293, sample_cdf :295-328).  Exposed one by one for parity tests;
 * nerfart_volsdf_fine_sample chains them. */
int nerfart_volsdf_first_check(int n_rays, int n, int cap, int n_final, float eps, float alpha_net, float beta_net,
                               const float* dA, const float* sA, const float* u_final, int u_final_stride,
                               float beta_plus0_denom, const float* far, float far_s, float* d_fine, float* beta_plus,
                               float* beta_map, float* iter_usage, int* act_out, int* act_count, void* stream);
int nerfart_volsdf_upsample(int n_active, int n, int cap, int n_up, const float* dA, const float* sA, const int* act,
                            const float* beta_plus, const float* u_up, int clamp_bounds, float* d_new, void* stream);
int nerfart_volsdf_merge_check(int n_active, int n, int cap, int n_up, int n_final, int max_bisect, int it, float eps,
                               float alpha_net, float beta_net, const float* dA, const float* sA, float* dB, float* sB,
                               const int* act, const float* d_new, const float* s_new, const float* u_final,
                               int u_final_stride, float* d_fine, float* beta_plus, float* beta_map, float* iter_usage,
                               int* act_out, int* act_count, void* stream);
int nerfart_volsdf_finalize(int n_active, int n, int cap, int n_final, const float* dA, const float* sA, const int* act,
                            const float* u_final, int u_final_stride, const float* beta_plus, float* d_fine,
                            float* beta_map, float* iter_usage, void* stream);
long long nerfart_volsdf_sampler_workspace_bytes(int n_rays, int n_init, int n_up, int n_final, int max_iter);
int nerfart_volsdf_fine_sample(const float* surf_blob, int precision, const float* rays_o, const float* rays_dn, int n_rays,
                               const float* near, const float* far, float near_s, float far_s, float R_bg,
                               float alpha_net, float beta_net, float eps, int n_init, int n_up, int n_final,
                               int max_iter, int max_bisect, const float* t_init_dev, const float* u_up_dev,
                               const float* u_final_dev, int u_final_per_ray, float* d_fine, float* beta_map,
                               float* iter_usage, void* workspace, long long workspace_bytes, void* stream);
/* t_init_dev / u_up_dev / u_final_dev: device copies of torch.linspace(0,1,n) for n = n_init, n_up+2, n_final
 * (the reference's tables, volsdf.py:483, rend_util.py:269,304); all three NULL -> nerfart_linspace is used.
 * u_final_per_ray != 0 (perturb=True: sample_cdf(det=False), rend_util.py:306-307): u_final_dev is [n_rays, n_final], the
 * caller's uniform random numbers, a row per ray (u_final_stride = n_final in the stage entry points; 0 = shared table). */


/* GUARDED Algorithm 1 (ABI 4): the SDF queries on (surf_blob, precision) - the cheap arithmetic, e.g. precision 4 - with every ray whose outcome hangs
 * on a marginal threshold decision sampled AGAIN, from its first query on, on (esc_blob, esc_precision):
 *   (i)  a ray whose max error bound lies within guard * eps of eps at a convergence check (`max B > eps`, volsdf.py:162-163, :240-242);
 *   (ii) a ray still active after the last round (volsdf.py:294-300: it is sampled with its last bisected beta+ - the rays ANY change of rounding
 *        moves, the reference's own on another machine included: DESIGN.md 2).
 * Rays are independent (volsdf.py:112), so the escalated rays' d_fine / beta_map / iter_usage are bit-identical to a run of all rays on esc_blob; the
 * others made every decision with a margin.  *n_escalated (host int, may be NULL): how many rays ran twice.  guard <= 0 or esc_blob NULL:
 * nerfart_volsdf_fine_sample.  Same workspace (nerfart_volsdf_sampler_workspace_bytes). */
int nerfart_volsdf_fine_sample_guarded(const float* surf_blob, int precision, const float* esc_blob, int esc_precision, float guard,
                                       const float* rays_o, const float* rays_dn, int n_rays,
                                       const float* near, const float* far, float near_s, float far_s, float R_bg,
                                       float alpha_net, float beta_net, float eps, int n_init, int n_up, int n_final,
                                       int max_iter, int max_bisect, const float* t_init_dev, const float* u_up_dev,
                                       const float* u_final_dev, int u_final_per_ray, float* d_fine, float* beta_map,
                                       float* iter_usage, int* n_escalated, void* workspace, long long workspace_bytes, void* stream);

/* (ABI 5) The same with a third escalation rule: late_round > 0 - a ray still active after up-sampling round `late_round` is escalated there (its remaining
 * rounds would each start from a 10-step bisection for beta+, volsdf.py:266-275, whose threshold decisions feed the next round's sampling density: the
 * branch-sensitive rays of Algorithm 1) instead of being carried through them on the cheap arithmetic.  late_round = 0: nerfart_volsdf_fine_sample_guarded. */
int nerfart_volsdf_fine_sample_guarded2(const float* surf_blob, int precision, const float* esc_blob, int esc_precision, float guard, int late_round,
                                        const float* rays_o, const float* rays_dn, int n_rays,
                                        const float* near, const float* far, float near_s, float far_s, float R_bg,
                                        float alpha_net, float beta_net, float eps, int n_init, int n_up, int n_final,
                                        int max_iter, int max_bisect, const float* t_init_dev, const float* u_up_dev,
                                        const float* u_final_dev, int u_final_per_ray, float* d_fine, float* beta_map,
                                        float* iter_usage, int* n_escalated, void* workspace, long long workspace_bytes, void* stream);

/* out[r] = sort(cat(a[r, :na], b[r, :nb]))  (volsdf.py:501-502) */
int nerfart_sort_concat(int n_rays, const float* a, int na, int a_stride, const float* b, int nb, int b_stride,
                        float* out, int out_stride, void* stream);

/* sdf_to_sigma + ray integration (volsdf.py:34-53, :544-576).  normals / sigma_out / p_out / tau_out may be NULL. */
int nerfart_volsdf_composite(int n_rays, int P, const float* d_all, const float* sdf, const float* radiance,
                             const float* nabla, float alpha, float beta, int white_bkgd, float* rgb, float* depth,
                             float* acc, float* normals, float* sigma_out, float* p_out, float* tau_out, void* stream);

/* Backward of nerfart_volsdf_composite w.r.t. the rgb output (volsdf.py:766) and, optionally, the opacity output acc
 * (g_acc [R] or NULL: the cotangent of mask_volume, the mask BCE of the reconstruction objective, neus.py:600-603):
 * g_rgb [R,3] -> g_sdf [R,P] (through sdf_to_sigma, volsdf.py:34-53), g_rad [R,P,3], and
 * g_alpha_beta[2] += (d loss / d alpha, d loss / d beta) (accumulated with atomics: zero it first; may be NULL).
 * What `rgb.backward(gradient)` (volsdf.py:766) does to the per-ray stage; first hand-written piece of B1 "bwd". */
int nerfart_volsdf_composite_bwd(int n_rays, int P, const float* d_all, const float* sdf, const float* radiance, float alpha,
                                 float beta, int white_bkgd, const float* g_rgb, const float* g_acc, float* g_sdf, float* g_rad,
                                 float* g_alpha_beta, void* stream);

/* The same for NeuS (neus.py:29-78, :373-395): sdf [R,P] at the samples, radiance [R,P-1,3] at the mid-points, s =
 * exp(ln_s * speed_factor) -> g_sdf [R,P], g_rad_mid [R,P-1,3], g_s[0] += d loss / d s. */
int nerfart_neus_composite_bwd(int n_rays, int P, const float* sdf, const float* rad_mid, float s, int white_bkgd, const float* g_rgb,
                               const float* g_acc, float* g_sdf, float* g_rad_mid, float* g_s, void* stream);

/* ---- B1: VolSDF volume_render (volsdf.py:389-615) for one chunk of rays (rays_d un-normalised). */
long long nerfart_volsdf_render_workspace_bytes(int n_rays, int n_samples, int n_importance, int max_upsample_steps,
                                                int k3_rays_chunk);
int nerfart_volsdf_render_fwd(const float* surf_blob, const float* rad_blob, int precision, int view_tiles, const float* rays_o,
                              const float* rays_d, int n_rays, float near_s, float far_s, float R_bg, float alpha,
                              float beta, float eps, int n_samples, int n_importance, int max_upsample_steps,
                              int max_bisection_steps, int white_bkgd, int k3_rays_chunk, const float* t_coarse_dev,
                              const float* t_init_dev, const float* u_up_dev, const float* u_final_dev, int u_final_per_ray,
                              float* rgb, float* depth, float* acc, float* normals, float* d_all_out, float* sdf_out, float* nabla_out,
                              float* radiance_out, float* sigma_out, float* p_out, float* tau_out, float* beta_map_out,
                              float* iter_usage_out, void* workspace, long long workspace_bytes, void* stream);

/* The same with Algorithm 1's SDF queries (the no-gradient sampling stage, volsdf.py:479: 512 (1 + rounds) queries per ray, 70 % of a frame) on
 * their OWN blob and precision - e.g. precision 4 (the 2-MFMA kernels) for the sampler and precision 1 for the 192 final samples, whose sdf / nabla /
 * radiance / compositing produce every number that reaches a pixel (+15 % frame rate inside the shipped mode's pixel budgets).  The
 * workspace is nerfart_volsdf_render_workspace_bytes'.  nerfart_volsdf_render_fwd = this with (sampler_blob, sampler_precision) = (surf_blob, precision). */
int nerfart_volsdf_render_mixed_fwd(const float* surf_blob, const float* rad_blob, int precision, const float* sampler_blob, int sampler_precision,
                                    int view_tiles, const float* rays_o, const float* rays_d, int n_rays, float near_s, float far_s, float R_bg, float alpha,
                                    float beta, float eps, int n_samples, int n_importance, int max_upsample_steps,
                                    int max_bisection_steps, int white_bkgd, int k3_rays_chunk, const float* t_coarse_dev,
                                    const float* t_init_dev, const float* u_up_dev, const float* u_final_dev, int u_final_per_ray,
                                    float* rgb, float* depth, float* acc, float* normals, float* d_all_out, float* sdf_out, float* nabla_out,
                                    float* radiance_out, float* sigma_out, float* p_out, float* tau_out, float* beta_map_out,
                                    float* iter_usage_out, void* workspace, long long workspace_bytes, void* stream);

/* Every stage on its own blob / precision (ABI 4; what the shipped `mixed` mode calls): Algorithm 1 on (sampler_blob, sampler_precision), GUARDED by
 * sampler_guard (> 0: nerfart_volsdf_fine_sample_guarded with (surf_blob, precision) as the escalation arithmetic; <= 0: off); sdf + nabla of the 192 final
 * samples on (surf_blob, precision); the radiance net there on (rad_blob, rad_precision); compositing in fp32.  *n_escalated: host int or NULL.
 * nerfart_volsdf_render_mixed_fwd = this with rad_precision = precision and sampler_guard = 0. */
int nerfart_volsdf_render_staged_fwd(const float* surf_blob, int precision, const float* rad_blob, int rad_precision, const float* sampler_blob,
                                     int sampler_precision, float sampler_guard, int view_tiles, const float* rays_o, const float* rays_d, int n_rays,
                                     float near_s, float far_s, float R_bg, float alpha, float beta, float eps, int n_samples, int n_importance,
                                     int max_upsample_steps, int max_bisection_steps, int white_bkgd, int k3_rays_chunk, const float* t_coarse_dev,
                                     const float* t_init_dev, const float* u_up_dev, const float* u_final_dev, int u_final_per_ray,
                                     float* rgb, float* depth, float* acc, float* normals, float* d_all_out, float* sdf_out, float* nabla_out,
                                     float* radiance_out, float* sigma_out, float* p_out, float* tau_out, float* beta_map_out,
                                     float* iter_usage_out, int* n_escalated, void* workspace, long long workspace_bytes, void* stream);

/* (ABI 5) nerfart_volsdf_render_staged_fwd with the sampler's late_round (nerfart_volsdf_fine_sample_guarded2); 0 = that function. */
int nerfart_volsdf_render_staged2_fwd(const float* surf_blob, int precision, const float* rad_blob, int rad_precision, const float* sampler_blob,
                                      int sampler_precision, float sampler_guard, int sampler_late_round, int view_tiles, const float* rays_o, const float* rays_d,
                                      int n_rays, float near_s, float far_s, float R_bg, float alpha, float beta, float eps, int n_samples, int n_importance,
                                      int max_upsample_steps, int max_bisection_steps, int white_bkgd, int k3_rays_chunk, const float* t_coarse_dev,
                                      const float* t_init_dev, const float* u_up_dev, const float* u_final_dev, int u_final_per_ray,
                                      float* rgb, float* depth, float* acc, float* normals, float* d_all_out, float* sdf_out, float* nabla_out,
                                      float* radiance_out, float* sigma_out, float* p_out, float* tau_out, float* beta_map_out,
                                      float* iter_usage_out, int* n_escalated, void* workspace, long long workspace_bytes, void* stream);

/* ---- NeuS (models/frameworks/neus.py): up-sampling 'official_solution' :275-303 ('direct_use' / 'direct_more' :242-269 below), helpers :29-78,
 * volume_render :142-424; near_far_from_sphere utils/rend_util.py:168-186. */
int nerfart_near_far_from_sphere(const float* rays_o, const float* rays_dn, int n_rays, float r, float* near,
                                 float* far, void* stream);
int nerfart_neus_upsample_step(int n_rays, int n, int cap, int n_new, float inv_s, const float* d, const float* sdf,
                               const float* u_new, int u_new_stride, float* d_new, void* stream);
int nerfart_merge_sorted_pairs(int n_rays, int n, int cap, int n_new, float* d, float* sdf, const float* d_new,
                               const float* s_new, void* stream);
int nerfart_neus_composite(int n_rays, int P, const float* d_all, const float* sdf, const float* radiance_mid,
                           const float* nabla, float s, int white_bkgd, float* rgb, float* depth, float* acc,
                           float* normals, float* cdf_out, float* alpha_out, float* w_out, float* d_mid_out,
                           void* stream);
long long nerfart_neus_render_workspace_bytes(int n_rays, int n_samples, int n_importance, int k3_rays_chunk);
int nerfart_neus_render_fwd(const float* surf_blob, const float* rad_blob, int precision, int view_tiles, const float* rays_o,
                            const float* rays_d, int n_rays, float obj_bounding_radius, float s, int n_samples,
                            int n_importance, int n_upsample_iters, int white_bkgd, int k3_rays_chunk,
                            const float* t_coarse_dev, const float* u_new_dev, int u_new_per_ray, float* rgb,
                            float* depth, float* acc, float* normals, float* d_all_out, float* sdf_out,
                            float* nabla_out, float* radiance_out, float* cdf_out, float* alpha_out, float* w_out,
                            float* d_mid_out, void* workspace, long long workspace_bytes, void* stream);
/* u_new_per_ray != 0 (perturb=True: sample_pdf(det=False), rend_util.py:269-272): u_new_dev is [n_rays, n_importance], the
 * caller's uniform random numbers - up-sampling round i uses columns [i * n_new, (i + 1) * n_new); 0: the shared
 * linspace(0, 1, n_new) table (u_new_dev [n_new] or NULL). */

/* ABI 3: the other two up-sampling algorithms of neus.py (:242-269; `model.upsample_algo` in the YAML, :735) behind the same renderer.
 *   upsample_algo 0 'official_solution' (= nerfart_neus_render_fwd), 1 'direct_use' (:242-255: all n_importance fine samples at once from the
 *   coarse samples' own visibility weights sdf_to_w(sdf, 1 / fixed_s_recp), :47-63), 2 'direct_more' (:259-269: the same from n_nograd_samples
 *   evenly spaced no-gradient samples).  For 1 and 2 n_upsample_iters is ignored and the uniform numbers are ONE table of n_importance values
 *   (u_new_per_ray == 0: u_new_dev [n_importance] = linspace(0, 1, n_importance), or NULL) or [n_rays, n_importance].
 *   t_nograd_dev: linspace(0, 1, n_nograd_samples) on the device or NULL (built inside: one stream synchronisation).
 *   nerfart_neus_direct_upsample_step: one inversion (n bins with row stride cap -> n_new sorted samples per ray), the stage entry. */
int nerfart_neus_direct_upsample_step(int n_rays, int n, int cap, int n_new, float inv_s, const float* d, const float* sdf,
                                      const float* u_new, int u_new_stride, float* d_new, void* stream);
long long nerfart_neus_render_algo_workspace_bytes(int n_rays, int n_samples, int n_importance, int k3_rays_chunk, int upsample_algo,
                                                   int n_nograd_samples);
int nerfart_neus_render_algo_fwd(const float* surf_blob, const float* rad_blob, int precision, int view_tiles, const float* rays_o,
                                 const float* rays_d, int n_rays, float obj_bounding_radius, float s, int n_samples,
                                 int n_importance, int n_upsample_iters, int upsample_algo, int n_nograd_samples, float fixed_s_recp,
                                 int white_bkgd, int k3_rays_chunk,
                                 const float* t_coarse_dev, const float* t_nograd_dev, const float* u_new_dev, int u_new_per_ray, float* rgb,
                                 float* depth, float* acc, float* normals, float* d_all_out, float* sdf_out,
                                 float* nabla_out, float* radiance_out, float* cdf_out, float* alpha_out, float* w_out,
                                 float* d_mid_out, void* workspace, long long workspace_bytes, void* stream);

/* ---- B4: CLIP ViT-B/32 image encoder (third-party `clip`: `model.encode_image(img)` of `clip.load("ViT-B/32", device="cuda")`,
 * reference call sites criteria/clip_loss.py:204-216, contrastive_loss.py:110-114, patchnce_loss.py:124-128) and its
 * backward w.r.t. the image (what autograd does there when the style loss is back-propagated, volsdf.py:912-915; the CLIP
 * weights are frozen).  csrc/clip_vit.hip: fp16 weights / GEMM operands on v_mfma_f32_32x32x16_f16, fp32 accumulation,
 * LayerNorm / softmax / residual stream in fp32.
 *   blob      : the `visual.*` parameters packed by nerfart_amd/clip_native.py in the section order
 *               nerfart_clip_vitb32_blob_layout() reports (offsets[203] in bytes, last = total size; returns the total):
 *               fp16 matrices, each stored once (0 conv1, 2 + 8 l + {0, 2, 4, 6} the four linear maps of block l, 99 proj; the odd
 *               sections, once the transposed copies, are empty: the backward GEMMs read the forward matrices in place), then
 *               fp32 vectors (100 class / positional embeddings, LayerNorm parameters, biases) - list in csrc/clip_vit.hip.
 *   blob_bytes: the size of the caller's blob; must equal the layout's total, so a blob packed to an older layout (ABI 1 stored
 *               transposed copies too) is rejected instead of read.
 *   img       : [B, 3, 224, 224] fp32, already resized and normalised (the reference's `preprocess`).
 *   feat_out  : [B, 512] fp32 (un-normalised features, as encode_image returns them).
 *   workspace : nerfart_clip_vitb32_workspace_bytes(B, keep_for_bwd) bytes of device memory.  With keep_for_bwd != 0 the
 *               forward leaves the per-block activations there and nerfart_clip_vitb32_image_bwd(g_feat [B,512] ->
 *               g_img [B,3,224,224]) must be given the SAME workspace, untouched; several forwards may be outstanding,
 *               each with its own workspace. */
long long nerfart_clip_vitb32_blob_layout(long long* offsets);
long long nerfart_clip_vitb32_workspace_bytes(int B, int keep_for_bwd);
int nerfart_clip_vitb32_image_fwd(const void* blob, long long blob_bytes, const float* img, int B, float* feat_out, int keep_for_bwd, void* workspace,
                                  long long workspace_bytes, void* stream);
int nerfart_clip_vitb32_image_bwd(const void* blob, long long blob_bytes, int B, const float* g_feat, float* g_img, void* workspace, long long workspace_bytes,
                                  void* stream);
/* ---- image side of the style losses (rows a20-a22; csrc/style_heads.hip).
 * nerfart_resample_fwd: one stage of the reference's torchvision `preprocess` chains (criteria/clip_loss.py:166-168,
 * contrastive_loss.py:98-101, patchnce_loss.py:98-117, :184-215) as a gather: the source image(s) src [n_src, C, Hs, Ws] (n_src = 1:
 * shared by all outputs, else N) sit at (pad_t, pad_l) inside a zero canvas Hp x Wp (ZeroPad2d), the canvas is resampled to
 * Hr x Wr (mode 1 = bicubic A = -0.75, 0 = bilinear; align_corners = False, no antialias), output image n [C, Ho, Wo] is the
 * window of it at crop_yx[n] = (y0, x0) (int32 [N, 2] on the device; NULL = (0, 0)), then v * affine[c] + affine[C + c]
 * (affine NULL = identity: (x + 1) / 2 and Normalize(mean, std) fold into it).  src_win (int32 [N, 4] = y, x, h, w on the device, or
 * NULL): output n resamples only that sub-window of its source, taps clamping at the window's edges - crop-then-resize, the
 * 112 x 112 -> 224 x 224 patches of patchnce_loss.py:213-215 (the canvas is then the window: no padding, Hp / Wp ignored).
 * nerfart_resample_bwd accumulates J^T g_dst into g_src (fp32 atomics; zero it first). */
int nerfart_resample_fwd(const float* src, int n_src, int C, int Hs, int Ws, int pad_t, int pad_l, int Hp, int Wp, int Hr, int Wr, int mode,
                         const int* crop_yx, const int* src_win, const float* affine, float* dst, int N, int Ho, int Wo, void* stream);
int nerfart_resample_bwd(const float* g_dst, int n_src, int C, int Hs, int Ws, int pad_t, int pad_l, int Hp, int Wp, int Hr, int Wr, int mode,
                         const int* crop_yx, const int* src_win, const float* affine, float* g_src, int N, int Ho, int Wo, void* stream);
/* The three CLIP heads from image features feats [4 + P, 512] (rows: 0 directional prediction, 1 directional source, 2 contrastive
 * prediction, 3 contrastive source, 4.. PatchNCE crops) and cached unit text features: text_dir [512] (clip_loss.py:234-246),
 * t_tgt / t_con [T, 512] (templates of the target / of the contrastive negative prompt), t_neg [S, T, 512]:
 *   out4 = {w_dir L_dir + w_con L_con + w_nce L_nce, L_dir (clip_loss.py:244-254), L_con (contrastive_loss.py:146-153, margin),
 *           L_nce (patchnce_loss.py:153-173, temperature tau, summed over the P crops)},  g_feats [4 + P, 512] = d out4[0] / d feats
 * (rows 1 and 3 are constants of the loss: zero). */
int nerfart_clip_style_heads(const float* feats, int n_patches, const float* text_dir, const float* t_tgt, const float* t_con, const float* t_neg,
                             int n_neg, int n_templates, float w_dir, float w_con, float w_nce, float margin, float tau, float* out4,
                             float* g_feats, void* stream);

/* ---- surface renderer (models/ray_casting.py): per-ray stages around the SDF query B3 (the marching and
 * refinement points are evaluated with nerfart_sdf_fwd_rays).  Masks are uint8 [R].
 * nerfart_first_crossing: root_finding_surface_points' analysis of the marched values val [R, n_steps] at depths depth [R, n_steps]
 *   (ray_casting.py:79-126): mask = first sign change of (val - tau) exists & goes outside -> inside & the ray starts outside;
 *   bracket [R, 4] = (d_low, f_low, d_high, f_high) around it; d_pred = first secant estimate (1 where mask is false).
 * nerfart_secant_update: one run_secant_method iteration (ray_casting.py:15-29) given f_mid [R] = sdf at the current d_pred.
 * nerfart_root_finish: depth / point outputs with the reference's fill values (ray_casting.py:137-152): inf (fill_inf) or far where
 *   nothing was hit, 0 where the ray starts inside, pt = 1 where mask is false.
 * nerfart_sphere_trace_step: one iteration of sphere_tracing_surface_points (ray_casting.py:175-180). */
int nerfart_first_crossing(const float* val, const float* depth, int n_rays, int n_steps, float logit_tau, unsigned char* mask,
                           unsigned char* mask_sign_change, unsigned char* mask_start_outside, float* bracket, float* d_pred, void* stream);
int nerfart_secant_update(const float* f_mid, int n_rays, float logit_tau, const unsigned char* mask, float* bracket, float* d_pred, void* stream);
int nerfart_root_finish(const float* rays_o, const float* rays_dn, int n_rays, const unsigned char* mask, const unsigned char* mask_start_outside,
                        const float* d_pred, const float* far, float far_s, int fill_inf, float* d_out, float* pt_out, void* stream);
int nerfart_sphere_trace_step(const float* sdf, int n_rays, const float* far, float far_s, float* d, unsigned char* mask, void* stream);

/* ---- VGG16 perceptual term (criteria/perp_loss.py:9-57): torchvision vgg16.features[:16] (through relu3_3) as
 * implicit-GEMM 3 x 3 convolutions on v_mfma_f32_32x32x2_f32 (fp32 operands as the reference's net; csrc/vgg_conv.hip), L1
 * between prediction and target features.
 *   blob : nerfart_vgg16_blob_layout() sections (offsets[22] bytes; packed by nerfart_amd/vgg.py): per conv l = 0..6 forward
 *          weights, backward (tap-flipped, channel-transposed) weights, bias - all fp32; blob_bytes must equal the layout's total.
 *   img2 : [2, 3, H, W] fp32 = the ImageNet-normalised, resized prediction then target (perp_loss.py:41-45); H, W multiples of 4,
 *          H W / 16 a multiple of 64 (224 x 224 in the reference).
 * nerfart_vgg16_l1_fwd writes loss_out[0] (device) = mean |relu3_3(pred) - relu3_3(target)|; with keep_for_bwd the workspace
 * (nerfart_vgg16_workspace_bytes) keeps the activations and nerfart_vgg16_l1_bwd turns upstream[0] (device scalar, NULL = 1)
 * into g_img [1, 3, H, W] = d loss / d img2[0]. */
long long nerfart_vgg16_blob_layout(long long* offsets);
long long nerfart_vgg16_workspace_bytes(int H, int W, int keep_for_bwd);
int nerfart_vgg16_l1_fwd(const void* blob, long long blob_bytes, const float* img2, int H, int W, float* loss_out, int keep_for_bwd, void* workspace,
                         long long workspace_bytes, void* stream);
int nerfart_vgg16_l1_bwd(const void* blob, long long blob_bytes, int H, int W, const float* upstream, float* g_img, void* workspace, long long workspace_bytes, void* stream);

/* ---- weight-gradient reductions of pass 2 (row a19; what autograd accumulates through volsdf.py:759-770): for n_mats matrix
 * pairs, dW[m] [256, a_cols] fp32 = Z_m^T A_m over `rows` rows and, if cs != NULL, cs[m] [256] = column sums of the first cs_rows
 * rows of Z_m (the bias gradients).  Z_m [rows, 256] and A_m [rows, a_cols] (a_cols = 256 or 64) are bf16 row-major matrices
 * z_stride / a_stride BYTES apart (the point-major dumps of the backward kernels, read in place; a stride of 0 shares one
 * operand); 16-byte aligned.  accumulate != 0 adds to dW / cs.  workspace = nerfart_wgrad_workspace_bytes(n_mats, rows, a_cols)
 * bytes, caller owned (split-K partial results).  csrc/wgrad.hip: v_mfma_f32_32x32x16_bf16 on ds_read_b64_tr_b16 fragments. */
long long nerfart_wgrad_workspace_bytes(int n_mats, long long rows, int a_cols);
int nerfart_wgrad_bf16(const void* Z, long long z_stride, const void* A, long long a_stride, int n_mats, long long rows, int a_cols,
                       long long cs_rows, float* dW, float* cs, int accumulate, void* workspace, long long workspace_bytes, void* stream);

/* ---- per-point glue of pass 2 (row a19; csrc/pass2_operands.hip): what autograd does between the network calls of
 * volsdf.py:759-770, one pass each.
 *   nerfart_ray_points: pts[R P,3] = rays_o + rays_dn * depth[R,P] (volsdf.py:503-506), view[R P,3] = rays_dn per sample (NULL: skip).
 *   nerfart_volsdf_pass2_cotangents: from the compositor's g_sdf[R P], the radiance net's g_n[R P,3] (+ g_n_extra; either may be NULL = zero) and
 *     pass 1's pts / sdf / nabla: sbar[R P] = g_sdf where the sphere clamp sdf = min(net, R_bg - |x|) kept the net (volsdf.py:97-100,
 *     else 0; R_bg <= 0: no sphere, NeuS), nbar[R P,3] = g_n + the eikonal term's gradient w 2 (|n| - 1) n / (|n| N) with N = the points of the ray's reference
 *     patch (eik_group_rays rays per patch inside this launch, the last one ragged; <= 0: one patch), eik_ray[R] = each ray's share
 *     of w * mean_patch((|n| - 1)^2) (their sum = the sum of the patches' eikonal losses).  w_eikonal = 0: no eikonal term.
 *   nerfart_wgrad_operand_*: the narrow (a_cols = 64) operands of nerfart_wgrad_bf16 from fp32 data: hi parts bf16(x) in columns
 *     0..c-1 and, when c <= 32, lo parts bf16(x - hi) in columns 32..32+c-1 (add the two halves of dW); rows M.. zero.
 *       _embed_pair: [2 rows_pad, 64]: embed(pts) (models/base.py:46-64, multires < 0: identity) in rows 0.., its tangent along dir in
 *                    rows rows_pad.. (the input side of SDF layers 0 and 4 for the stacked [value; tangent] dumps)
 *       _inputs:     [rows_pad, 64]: [embed(x, multires_x) | embed(view, multires_view) | normals] (radiance layer 0)
 *       _rgb_delta:  [rows_pad, 64]: g_rgb rgb (1 - rgb) (the output sigmoid's delta, 3 columns); also in fp32 to d4[M,3] (NULL: skip)
 *                    and its sums over each block of 32 rows to block_sums[ceil(rows_pad / 32), 3] (their sum = the bias gradient)
 *       _sbar_ones:  [2 rows_pad, 64]: sbar (1 column) in rows 0.., 1 in rows rows_pad.. (the sdf row of the last SDF layer) */
int nerfart_ray_points(const float* rays_o, const float* rays_dn, const float* depth, long long n_rays, int P, float* pts, float* view,
                       void* stream);
int nerfart_volsdf_pass2_cotangents(const float* pts, const float* sdf, const float* g_sdf, const float* nabla, const float* g_n,
                                    const float* g_n_extra, long long n_rays, int P, float R_bg, float w_eikonal,
                                    long long eik_group_rays, float* sbar, float* nbar, float* eik_ray, void* stream);
int nerfart_wgrad_operand_embed_pair(const float* pts, const float* dir, long long M, long long rows_pad, int multires, void* out,
                                     void* stream);
int nerfart_wgrad_operand_inputs(const float* x, int multires_x, const float* view, int multires_view, const float* normals, long long M,
                                 long long rows_pad, void* out, void* stream);
int nerfart_wgrad_operand_rgb_delta(const float* rgb, const float* g_rgb, long long M, long long rows_pad, void* out, float* d4,
                                    float* block_sums, void* stream);
int nerfart_wgrad_operand_sbar_ones(const float* sbar, long long M, long long rows_pad, void* out, void* stream);

/* ---- B1 "bwd": the RAY-LEVEL backward of the renderer (csrc/render_backward.hip).  What `rgb.backward(gradient)` +
 * `eikonal.backward()` through render_fn do for one patch of rays in the reference (models/frameworks/volsdf.py:759-770,
 * neus.py:520-576) is one call: the whole pass-2 kernel sequence above (points, [SDF + nabla + h7], radiance forward with dumps,
 * compositor backward, radiance backward, cotangents, second-order SDF sweeps, weight-gradient reductions) on the caller's stream,
 * out of one caller-owned workspace.  Split-bf16 blobs (precision 1) only.
 *
 *   raw       : the RAW parameter-gradient buffer, nerfart_pass2_raw_layout(offsets[14]) floats (offsets = float offset of each
 *               section, last = total; returns the total): the fp32 results of the reductions in the kernels' unit order, bias column
 *               sums, and 4 scalars (section 12: d loss / d alpha, d loss / d beta [VolSDF], d loss / d s [NeuS], the sum of the
 *               patches' eikonal losses).  Every call ACCUMULATES into it: zero it once per optimiser step, call once per launch group
 *               (n_rays * P <= 2^21), then nerfart_fold_weight_grads once.  Everything downstream is linear in it, so ranks may also
 *               all-reduce the raw buffer instead of the parameter gradients.
 *   rays_d    : un-normalised, as nerfart_*_render_fwd takes them; d_all [n_rays, P]: the sample depths pass 1 drew (d_all_out of
 *               the forward; sampling carries no gradient, volsdf.py:479).
 *   g_rgb     : d loss / d rgb [n_rays, 3]; g_acc [n_rays] or NULL: d loss / d mask_volume (the mask BCE of neus.py:600-603);
 *               g_n_extra [n_rays, P, 3] or NULL: a further cotangent of the nablas (VolSDF reconstruction objective, volsdf.py:803-806).
 *   *_state   : what pass 1 computed at these samples (sdf_out / nabla_out of the forward, and for VolSDF the layer-7 activation h7
 *               [n_rays P, 256] of nerfart_sdf_nabla_fwd) - same weights, identical values; all NULL: recomputed here.
 *   w_eikonal : weight of mean((|nabla| - 1)^2) (0: no eikonal term); eik_group_rays: the rays are several reference patches of
 *               that many rays in one launch, each with its OWN mean (<= 0: one patch).
 *   train_radiance == 0: the radiance net is frozen (neus.py:455-456): only the SDF net's gradients (incl. the geometry-feature
 *               rows of its last layer) are produced.
 * nerfart_sdf_param_bwd: the SDF net's share on its own - parameter gradients of  sbar . sdf + hbar7 . h7 + nbar . grad_x sdf  at
 *   pts [M, 3] (sbar [M] / hbar7 [M, 256] may be NULL = zero; M <= 2^21): the free eikonal points of volsdf.py:799-806.
 * nerfart_fold_weight_grads: raw -> `folded`, the gradients of the FOLDED weights W = g v / |v| and of the biases in the reference's
 *   feature order, laid out by nerfart_folded_grads_layout(multires, multires_view, offsets[29]): (dW_l [out_l, in_l], db_l [out_l])
 *   for the SDF net's layers 0..8, then the radiance net's 0..4; offsets in floats, last = total (returned).  multires = the SDF
 *   net's embed_multires (6), multires_view = the radiance net's embed_multires_view (-1 | 4).
 * nerfart_weight_norm_bwd: nn.utils.weight_norm's chain rule for one layer (models/base.py:226-227): dW [out, in] ->
 *   g_weight_v [out, in], g_weight_g [out] (either may be NULL; accumulate != 0 adds to them). */
long long nerfart_pass2_raw_layout(long long* offsets);
long long nerfart_volsdf_render_bwd_workspace_bytes(int n_rays, int P, int have_state);
int nerfart_volsdf_render_bwd(const float* surf_blob, const float* rad_blob, int view_tiles, int multires, const float* rays_o, const float* rays_d,
                              int n_rays, int P, const float* d_all, const float* g_rgb, const float* g_acc, const float* g_n_extra,
                              const float* sdf_state, const float* nabla_state, const float* h7_state, float R_bg, float alpha, float beta,
                              int white_bkgd, float w_eikonal, int eik_group_rays, int train_radiance, float* raw, void* workspace,
                              long long workspace_bytes, void* stream);
long long nerfart_neus_render_bwd_workspace_bytes(int n_rays, int P, int have_state);
int nerfart_neus_render_bwd(const float* surf_blob, const float* rad_blob, int view_tiles, int multires, const float* rays_o, const float* rays_d,
                            int n_rays, int P, const float* d_all, const float* g_rgb, const float* g_acc, const float* sdf_state,
                            const float* nabla_state, float s, int white_bkgd, float w_eikonal, int eik_group_rays, int train_radiance, float* raw,
                            void* workspace, long long workspace_bytes, void* stream);
long long nerfart_sdf_param_bwd_workspace_bytes(long long M);
int nerfart_sdf_param_bwd(const float* surf_blob, int multires, const float* pts, long long M, const float* sbar, const float* hbar7, const float* nbar,
                          float* raw, void* workspace, long long workspace_bytes, void* stream);
/* the radiance net's share on its own (B2 "bwd", parameter level): RadianceNet.forward (models/base.py:372-391) on [pts, view, nabla,
 * W8[1:] h7 + b8[1:]] with its backward - rgb_out [M,3], g_h7_out [M,256], g_n_out [M,3] (each may be NULL) and, accumulated into
 * raw, the gradients of the radiance net (train_radiance != 0) and of the geometry-feature rows of the last SDF layer. */
long long nerfart_radiance_param_bwd_workspace_bytes(long long M);
int nerfart_radiance_param_bwd(const float* rad_blob, int view_tiles, const float* pts, const float* view, const float* nabla, const float* h7,
                               long long M, const float* g_rgb, float* rgb_out, float* g_h7_out, float* g_n_out, int train_radiance, float* raw,
                               void* workspace, long long workspace_bytes, void* stream);
long long nerfart_folded_grads_layout(int multires, int multires_view, long long* offsets);
int nerfart_fold_weight_grads(const float* raw, int multires, int multires_view, float* folded, void* stream);
int nerfart_weight_norm_bwd(const float* dW, const float* weight_v, const float* weight_g, int out_features, int in_features, float* g_weight_v,
                            float* g_weight_g, int accumulate, void* stream);

/* ---- B5: checkpoint -> blobs (reference render.py:266-267 `model.load_state_dict(torch.load(p)['model'])`, models/base.py:226-227
 * `nn.utils.weight_norm`: the state dict holds weight_g [out, 1], weight_v [out, in], bias [out] per layer).  csrc/pack_blob.hip turns
 * those tensors into the blobs every entry point above reads, entirely on the device and on the caller's stream: the weight_norm fold
 * w = g * v / ||v||_row, the permutation into the kernels' fragment order (a closed form the library owns; tests/test_pack_plan.py holds it
 * equal, entry for entry, to the numpy plans of nerfart_amd/packing.py that the CPU emulation walks) and, for the split programs, hi = rne(w),
 * lo = rne(w - hi) in bf16 (precision 1) or fp16 (precision 4).  Call once per weight update (two launches per blob).
 *   precision       : the C-ABI precision the blob is for (5: nerfart_pack_surface_blob only, see the precision list above): 0 fp32, 1 split bf16 ("bf16x3"; also what every training entry point reads),
 *                     4 fp16 hi + lo ("fp16x2"; the sampler blob of nerfart_volsdf_render_mixed_fwd)
 *   weight_g/_v/bias: HOST arrays of DEVICE pointers, one per layer: `implicit_surface.surface_fc_layers.{0..8}.*` (W 256, D 8, skips [4],
 *                     embed_multires 6, W_geo_feat 256 - the four shipped configs) resp. `radiance_net.layers.{0..4}.*` (W 256, D 4;
 *                     view_tiles 1: raw view directions, input 265; 3: embed_multires_view 4, input 289); surf8_*: the SDF net's LAST
 *                     layer, whose rows 1..256 (the geometry feature) the radiance kernels evaluate
 *   blob_out        : nerfart_*_blob_floats(precision, ...) floats (0 = bad arguments, see nerfart_last_error)
 *   workspace       : nerfart_pack_workspace_bytes() bytes of scratch (the rows' 1 / ||v||)
 * nerfart_pack_plan_debug is HOST-only (tests): the layout as plain gather tables - sizes[4] = {chunk elements, aux elements, total floats,
 * chunks}; cindex / cmul (second factor) / aindex: flat indices into [tensors in state-dict order w0, b0, w1, ... | 0.0 | 1.0]; cscale per
 * chunk element; header: the 512 header words.  Any table may be NULL. */
long long nerfart_surface_blob_floats(int precision, int multires);
long long nerfart_radiance_blob_floats(int precision, int view_tiles);
long long nerfart_pack_workspace_bytes(void);
/* (ABI 4) The geometry feature of ImplicitSurface.forward(x, return_h=True) / forward_with_nablas (models/base.py:243-282): rows 1..256 of the SDF net's
 * LAST linear layer on the layer-7 activations, feat = W8[1:] h7 + b8[1:] with W8 = weight_g * weight_v / ||weight_v|| folded on the device (ATen's
 * summation order) and the product on v_mfma_f32_32x32x2_f32 (exact fp32 products; csrc/geo_feature.hip).  weight_g [257], weight_v [257, 256],
 * bias [257]: `implicit_surface.surface_fc_layers.8.*`; h7 [M, 256]: nerfart_sdf_nabla_fwd's output; feat_out [M, 256].  The render path never calls
 * it (the radiance kernels evaluate these rows from their own blob): it serves the reference's other consumers of the callable (boundary B3). */
long long nerfart_geometry_feature_workspace_bytes(void);
int nerfart_geometry_feature(const float* weight_g, const float* weight_v, const float* bias, const float* h7, long long M, float* feat_out,
                             void* workspace, long long workspace_bytes, void* stream);

/* (ABI 4) The architecture the packers index their pointer tables with, as data: radiance == 0 -> the SDF net's 9 layers for embed_multires = arg (6);
 * != 0 -> the radiance net's 5 layers for view_tiles = arg (1 | 3).  rows[l] x cols[l] = weight_v[l]'s shape (weight_g [rows, 1], bias [rows]); the
 * return value is the layer count, 0 on a bad argument.  A host checks a checkpoint against it BEFORE calling nerfart_pack_*_blob (any other W / D /
 * skips / W_geo_feat would be read out of bounds). */
int nerfart_pack_layer_dims(int radiance, int arg, int* rows, int* cols);
int nerfart_pack_surface_blob(int precision, int multires, const float* const* weight_g, const float* const* weight_v, const float* const* bias,
                              float* blob_out, long long blob_floats, void* workspace, long long workspace_bytes, void* stream);
int nerfart_pack_radiance_blob(int precision, int view_tiles, const float* surf8_g, const float* surf8_v, const float* surf8_bias,
                               const float* const* weight_g, const float* const* weight_v, const float* const* bias, float* blob_out, long long blob_floats,
                               void* workspace, long long workspace_bytes, void* stream);
int nerfart_pack_plan_debug(int program, int view_tiles, int fp16, long long* sizes, int* header, int* cindex, int* cmul, float* cscale, int* aindex);

/* The packers of the CLIP / VGG blobs (ABI 3; the weights are frozen: once per run).  CLIP: `tensors` = HOST array of
 * nerfart_clip_vitb32_n_tensors() (= 152) DEVICE pointers to fp32 copies of the `visual.*` entries of the CLIP state dict, tensor i being
 * `visual.` + the name nerfart_clip_vitb32_tensor_name(i, buf, len) writes (its return value: the element count; 0 = bad index) - matrices are
 * stored fp16 (round to nearest even, like `clip.load(device="cuda")`'s weights), vectors fp32.  VGG16: weight[l] [Cout, Cin, 3, 3] / bias[l] of
 * torchvision's `features.{0, 2, 5, 7, 10, 12, 14}` (criteria/perp_loss.py:9-33), fp32 -> the forward / transposed-flipped backward / bias sections. */
int nerfart_clip_vitb32_n_tensors(void);
long long nerfart_clip_vitb32_tensor_name(int i, char* name_out, int name_len);
int nerfart_clip_vitb32_pack(const float* const* tensors, void* blob, long long blob_bytes, void* stream);
int nerfart_vgg16_pack(const float* const* weight, const float* const* bias, void* blob, long long blob_bytes, void* stream);

/* The GEMM kernel of the encoder on its own (tests): C[M,N] fp32 = A[M,K] fp16 . W[N,K]^T fp16; M, N, K multiples of 64. */
int nerfart_gemm_f16_nt(const void* A, const void* W, int M, int N, int K, float* C, void* stream);
/* ... and C[M,N] = A[M,K] . Wt[K,N] (second operand read with its reduction index as the row: the backward GEMMs on the forward
 * weight matrices, through ds_read_b64_tr_b16). */
int nerfart_gemm_f16_nn(const void* A, const void* Wt, int M, int N, int K, float* C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NERFART_HIP_H */
