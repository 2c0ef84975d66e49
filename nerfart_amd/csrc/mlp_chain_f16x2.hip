// mlp_chain_f16x2.hip - C-ABI precision 4 ("fp16x2"): the split-bf16 kernels' whole data flow compiled a second time with the 2-MFMA
// split (mlp_bf16_core.h, NERFART_F16X2): ONE fp16 activation term x fp16 hi + lo weight terms on v_mfma_f32_16x16x32_f16.
//
// Why it exists: the sustained rate of k_sdf_only_bf16 is set by the package power cap, i.e. by joules per
// product, and the only lever left is fewer matrix operations per product.  The 2-MFMA form had been rejected on a
// 512-ray maximum from a CPU emulation; this builds it for real so that it is measured with the statistics the shipped mode is
// held to (tools/parity_table.py: rays past 1e-3, max, p99.9, PSNR against the oracle and against the exact-fp32 frame) and timed
// (bench.py secondary.fp16x2).  It is NEVER the headline precision: 11-bit activations are TF32-class arithmetic.
//
// Only the three forward kernels the renderer needs are instantiated (K2 k_sdf_only, reverse-mode k_sdf_grad, k_radiance); the
// training kernels (fwd2 / bwd2 / radiance_bwd / dumps) stay split-bf16 only.  Blobs: packing.surface_plan_bf16(term="fp16") /
// radiance_plan_bf16(term="fp16") - same geometry, fp16 fragments, the reverse chunks WITHOUT the folded 1/65535 (see core).
#define NERFART_F16X2 1
#define b16 f16x2
#define sdf_bf16_v1 sdf_f16x2
#define sdf_nabla_bf16 sdf_nabla_fwdmode_f16x2
#define radiance_bf16 radiance_f16x2
#define radiance_dump_bytes radiance_dump_bytes_f16x2
#define radiance_fwd_dump_bf16 radiance_fwd_dump_f16x2
#define sdf_grad_ws_bytes sdf_grad_ws_bytes_f16x2
#define sdf_grad_bf16 sdf_grad_f16x2
#include "mlp_chain_bf16.hip"
#include "mlp_grad_bf16.hip"
