#!/usr/bin/env python
"""Timing ablation of k_sdf_only_w32 (csrc/mlp_k2_w32.hip): variant libraries with one component compiled out (results WRONG by
construction - this only attributes time).   python tools/archive/ablate_w32.py build   (here)  /  run   (on the GPU box)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "nerfart_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_ablate")
VARIANTS = {"full": [], "nodma": ["-DW32_NO_DMA"], "nomfma": ["-DW32_NO_MFMA"], "noepi": ["-DW32_NO_EPI"], "nolds": ["-DW32_NO_LDS"],
            "mfma_only": ["-DW32_NO_DMA", "-DW32_NO_EPI", "-DW32_NO_LDS"], "nomfma_nodma": ["-DW32_NO_MFMA", "-DW32_NO_DMA"],
            "dma_only": ["-DW32_NO_MFMA", "-DW32_NO_EPI", "-DW32_NO_LDS"]}


def build():
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(CSRC, "_build", f) for f in os.listdir(os.path.join(CSRC, "_build")) if f.endswith(".o") and f != "mlp_k2_w32.o"]
    for name, flags in VARIANTS.items():
        obj = os.path.join(OUT, f"w32_{name}.o")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-x", "hip"] + flags +
                              ["-c", os.path.join(CSRC, "mlp_k2_w32.hip"), "-o", obj])
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, f"libw32_{name}.so")] + objs + [obj])
        os.remove(obj)
        print("built", name, flush=True)


def run():
    res = {}
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from nerfart_amd import scene, hip
model, _, _ = scene.build_model("VolSDF", seed=0, beta=0.01, device="cuda", precision="bf16x3")
blob, _ = model.packed()
pts = (torch.rand(1 << 22, 3, generator=torch.Generator().manual_seed(0)) * 4 - 2).cuda()
hip.sdf_fwd(blob, pts, 3.0, precision=1); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): hip.sdf_fwd(blob, pts, 3.0, precision=1)
e1.record(); torch.cuda.synchronize()
print("MS", e0.elapsed_time(e1) / 10)
''' % ROOT
    for name in VARIANTS:
        env = dict(os.environ, NERFART_HIP_LIB=os.path.join(OUT, f"libw32_{name}.so"))
        env.pop("NERFART_K2", None)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        ms = [float(l.split()[1]) for l in r.stdout.splitlines() if l.startswith("MS")]
        res[name] = round(ms[0], 3) if ms else r.stderr[-300:]
        print(name, res[name], flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
