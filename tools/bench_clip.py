#!/usr/bin/env python
"""The hand-written CLIP ViT-B/32 image encoder (csrc/clip_vit.hip, gemm_f16.h) on its own: B = 16 images (the 4 + 12 of a
fine-tune step) forward + backward to the pixels, ms per call and TFLOP/s against the 2.5 PFLOP/s dense fp16 MFMA peak.
Algorithmic flops: 2 * 4.4 GMAC per image forward, the same again for the backward to the pixels (weights frozen:
no weight-gradient GEMMs)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nerfart_amd import clip_vit, clip_native

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = "cuda:0"
enc = clip_native.NativeImageEncoder(clip_vit.build_clip(dev, seed=0))
x = torch.randn(B, 3, 224, 224, device=dev, requires_grad=True)
cot = torch.randn(B, 512, device=dev)
def call():
    x.grad = None
    f = enc(x)
    f.backward(cot)
for _ in range(3):
    call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    call()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
# per image: patch embed 49 x 768 x 3072, 12 blocks of (50 x 768 x 2304 + 50 x 768 x 768 + 2 x 50 x 768 x 3072 + attention 2 x 12 x 50 x 50 x 64), proj 768 x 512
mac = 49 * 768 * 3072 + 12 * (50 * 768 * 2304 + 50 * 768 * 768 + 2 * 50 * 768 * 3072 + 2 * 12 * 50 * 50 * 64) + 768 * 512
flops = 2 * mac * B * 2
print(json.dumps({"B": B, "ms_fwd_bwd": round(ms, 3), "gflop_fwd_bwd": round(flops / 1e9, 1), "tflops": round(flops / ms / 1e9, 1),
                  "frac_of_2500": round(flops / ms / 1e9 / 2500, 4)}))
