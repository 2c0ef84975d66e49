#!/usr/bin/env python
"""Schedule sweep of k_sdf_only_w32 (csrc/mlp_k2_w32.hip): variant libraries that differ in -D macros (placement of the LDS-DMA piece
inside a double item, ...), each timed on 4 M points and checked against the default library's sdf.
   python tools/archive/sweep_w32.py build   (here)   /   python tools/archive/sweep_w32.py run   (on the GPU box)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "nerfart_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_ablate")
VARIANTS = {f"dma_at_{k}": [f"-DW32_DMA_AT={k}"] for k in range(12)}
VARIANTS.update(json.loads(os.environ.get("W32_EXTRA", "{}")))


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


CODE = r'''
import sys, torch
sys.path.insert(0, %r)
from nerfart_amd import scene, hip
model, _, _ = scene.build_model("VolSDF", seed=0, beta=0.01, device="cuda", precision="bf16x3")
blob, _ = model.packed()
pts = (torch.rand(1 << 22, 3, generator=torch.Generator().manual_seed(0)) * 4 - 2).cuda()
out = hip.sdf_fwd(blob, pts, 3.0, precision=1); torch.cuda.synchronize()
sdf = out[0] if isinstance(out, (tuple, list)) else out
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): hip.sdf_fwd(blob, pts, 3.0, precision=1)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10)
print("MS", best, "SUM", float(sdf.double().sum()), "ABS", float(sdf.double().abs().sum()))
'''


def run():
    res = {}
    code = CODE % ROOT
    todo = [("v1_default", None, "")] + [(n, os.path.join(OUT, f"libw32_{n}.so"), "w32") for n in VARIANTS]
    for name, lib, k2 in todo:
        env = dict(os.environ)
        env.pop("NERFART_K2", None)
        if lib:
            env["NERFART_HIP_LIB"] = lib
            env["NERFART_K2"] = k2
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("MS")]
        res[name] = line[0] if line else r.stderr[-300:]
        print(name, res[name], flush=True)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
