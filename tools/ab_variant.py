#!/usr/bin/env python
"""A/B of the split-bf16 MLP kernels against a variant library built with extra -D flags (e.g. -DNERFART_OLD_ITEM = the round-2
item form: separate wait / MFMA statements, asm relu): ms per launch of K2 (sdf only), K3a (sdf + nabla + h7) and K3b (radiance) on
the same points, and whether the outputs are bit-identical.
    python tools/ab_variant.py build NAME -DFLAG [-DFLAG ...]      (here; writes gpurun_ablate/libvar_NAME.so)
    python tools/ab_variant.py run NAME [NAME ...]                 (on the GPU box)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nerfart_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_ablate")
BF16_SOURCES = ["mlp_chain_bf16", "mlp_grad_bf16", "mlp_backward_bf16"]


def build(name, defs):
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(CSRC, "_build", f) for f in os.listdir(os.path.join(CSRC, "_build")) if f.endswith(".o") and f[:-2] not in BF16_SOURCES]
    mine = []
    for src in BF16_SOURCES:
        obj = os.path.join(OUT, f"{name}_{src}.o")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-x", "hip"] + defs +
                              ["-c", os.path.join(CSRC, src + ".hip"), "-o", obj])
        mine.append(obj)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, f"libvar_{name}.so")] + objs + mine)
    for o in mine:
        os.remove(o)
    print("built", name, defs)


CHILD = r'''
import sys, json, torch
sys.path.insert(0, %r)
from nerfart_amd import scene, hip
model, _, _ = scene.build_model("VolSDF", seed=0, beta=0.01, device="cuda", precision="bf16x3")
surf, rad = model.packed()
g = torch.Generator().manual_seed(0)
pts = (torch.rand(1 << 22, 3, generator=g) * 4 - 2).cuda()
view = torch.nn.functional.normalize(torch.randn(1 << 22, 3, generator=g), dim=-1).cuda()
def timed(fn, reps=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out
res = {}
res["k2_ms_per_4M"], sdf = timed(lambda: hip.sdf_fwd(surf, pts, 3.0, precision=1))
n3 = 3 << 19
res["k3a_ms_per_1.5M"], (s3, nab, h7) = timed(lambda: hip.sdf_nabla_fwd(surf, pts[:n3], 3.0, precision=1))
res["k3b_ms_per_1.5M"], rgb = timed(lambda: hip.radiance_fwd(rad, 1, pts[:n3], view[:n3], nab, h7, precision=1))
torch.save({"sdf": sdf.cpu(), "s3": s3.cpu(), "nab": nab.cpu(), "h7": h7[:65536].cpu(), "rgb": rgb.cpu()}, sys.argv[1])
print("RES", json.dumps({k: round(v, 4) for k, v in res.items()}))
''' % ROOT


def run(names):
    import torch
    outs = {}
    for name in ["main"] + names:
        env = dict(os.environ)
        if name != "main":
            env["NERFART_HIP_LIB"] = os.path.join(OUT, f"libvar_{name}.so")
        path = f"/tmp/ab_{name}.pt"
        r = subprocess.run([sys.executable, "-c", CHILD, path], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("RES")]
        print(name, line[0][4:] if line else r.stderr[-1500:])
        if line:
            outs[name] = torch.load(path)
    for name in names:
        if name in outs and "main" in outs:
            same = {k: bool(torch.equal(outs["main"][k], outs[name][k])) for k in outs["main"]}
            diff = {k: float((outs["main"][k] - outs[name][k]).abs().max()) for k in outs["main"]}
            print(json.dumps({"variant": name, "bit_identical_to_main": same, "max_abs_diff": diff}))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2], sys.argv[3:])
    else:
        run(sys.argv[2:])
