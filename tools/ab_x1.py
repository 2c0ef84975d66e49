#!/usr/bin/env python
"""A/B of the 1-MFMA K2 (csrc/mlp_chain_f16x1.hip, C-ABI precision 5) against variant libraries built with extra -D flags (the scheduling experiments of
mlp_bf16_core.h, re-asked for the kernel that is bound by its issue port instead of the power cap): ms per 4 M points and bit-identity of the sdf.
    python tools/ab_x1.py build NAME -DFLAG [...]     (here; gpurun_ablate/libx1_NAME.so)
    python tools/ab_x1.py run NAME [NAME ...]         (on the GPU box)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nerfart_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_ablate")


def build(name, defs):
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(CSRC, "_build", f) for f in os.listdir(os.path.join(CSRC, "_build")) if f.endswith(".o") and f != "mlp_chain_f16x1.o" and ".scan_generic" not in f]
    obj = os.path.join(OUT, f"x1_{name}.o")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-x", "hip"] + defs +
                          ["-c", os.path.join(CSRC, "mlp_chain_f16x1.hip"), "-o", obj])
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, f"libx1_{name}.so")] + objs + [obj])
    os.remove(obj)
    print("built", name, defs)


CHILD = r'''
import sys, json, torch
sys.path.insert(0, %r)
from nerfart_amd import scene, hip
model, _, _ = scene.build_model("VolSDF", seed=0, beta=0.01, device="cuda", precision="bf16x3")
model.set_sampler_precision("fp16x1", guard=0.05)
blob = model.packed_sampler()[0]
g = torch.Generator().manual_seed(0)
pts = ((torch.rand(1 << 22, 3, generator=g) * 2 - 1) * 1.5).cuda()
fn = lambda: hip.sdf_fwd(blob, pts, 3.0, precision=5)
fn(); torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6): out = fn()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 6)
torch.save(out.cpu(), sys.argv[1])
print("RES", json.dumps({"k2_x1_ms_per_4M": round(best, 4)}))
''' % ROOT


def run(names):
    import torch
    outs = {}
    for name in ["main"] + names + ["main"]:
        env = dict(os.environ)
        if name != "main":
            env["NERFART_HIP_LIB"] = os.path.join(OUT, f"libx1_{name}.so")
        path = f"/tmp/abx1_{name}.pt"
        r = subprocess.run([sys.executable, "-c", CHILD, path], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("RES")]
        print(name, line[0][4:] if line else r.stderr[-1500:], flush=True)
        if line:
            outs[name] = torch.load(path)
    for name in names:
        if name in outs and "main" in outs:
            print(json.dumps({"variant": name, "bit_identical_to_main": bool(torch.equal(outs["main"], outs[name])), "max_abs_diff": float((outs["main"] - outs[name]).abs().max())}))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2], sys.argv[3:])
    else:
        run(sys.argv[2:])
