// mlp_chain_f16x1.hip - C-ABI precision 5 ("fp16x1"): K2 (sdf only) with ONE matrix instruction per product - one fp16 activation term x one fp16
// weight term on v_mfma_f32_16x16x32_f16, fp32 accumulate - on the k-steps whose input unit is built from the previous layer's accumulators (55 of
// K2's 59); the ready-made input units (the positional encodings of layers 0 and 4) keep hi + lo terms on both sides and the three-term form, as in
// the 2-MFMA build.  Same data flow, same blob as precision 4 (the fp16 hi + lo fragments: the lo fragment of a 1-MFMA item is streamed but never
// read from LDS: half the fragment reads per item).
//
// What it is for: ONLY Algorithm 1's no-gradient SDF queries (volsdf.py:479) - 512 (1 + rounds) per ray, whose one product is WHERE the 64 fine
// samples sit - and only behind the guard of nerfart_volsdf_fine_sample_guarded (marginal and never-converged rays are sampled again on the model's own
// arithmetic).  A rounded weight is a FIXED, smooth perturbation of the SDF (~2^-12 relative per weight), not point-to-point noise: the hidden
// activations' 11 bits already set the sampler's resolution (tools/emul_sampler_precision.py: sdf error mean 1.55e-4 against the 2-MFMA form's 1.36e-4).
// No other entry point accepts precision 5: no value that reaches a pixel and no gradient is ever computed in it.
#define NERFART_F16X2 1
#define NERFART_F16X1 1
#define NERFART_K2_ONLY 1
#define b16 f16x1
#define sdf_bf16_v1 sdf_f16x1
#include "mlp_chain_bf16.hip"
