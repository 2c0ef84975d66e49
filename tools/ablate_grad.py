#!/usr/bin/env python
"""What the softplus' scratch round trip and the h7 hand-off cost k_sdf_grad_bf16 (csrc/mlp_grad_bf16.hip): a variant library without
the scratch loads / stores (results WRONG by construction - timing only) against the real one, with and without the h7 output.
    python tools/ablate_grad.py build   (here)   /   run   (on the GPU box)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nerfart_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_ablate")


VARIANTS = {
    "no_scratch": ["-DNERFART_ABLATE_SCRATCH"],
    "no_scratch_stores": ["-DNERFART_ABLATE_SCRATCH_ST"],
    "no_scratch_loads": ["-DNERFART_ABLATE_SCRATCH_LD"],
    "small_scratch": ["-DNERFART_EXP_SCRATCH_SMALL"],
    "late_store": ["-DNERFART_EXP_ST_LATE"],
    "late_store_no_loads": ["-DNERFART_EXP_ST_LATE", "-DNERFART_ABLATE_SCRATCH_LD"],
}


def build():
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(CSRC, "_build", f) for f in os.listdir(os.path.join(CSRC, "_build")) if f.endswith(".o") and f != "mlp_grad_bf16.o"]
    for name, defs in VARIANTS.items():
        obj = os.path.join(OUT, f"grad_{name}.o")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-x", "hip"] + defs +
                              ["-c", os.path.join(CSRC, "mlp_grad_bf16.hip"), "-o", obj])
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, f"libgrad_{name}.so")] + objs + [obj])
        os.remove(obj)
    print("built", list(VARIANTS))


def run():
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from nerfart_amd import scene, hip
model, _, _ = scene.build_model("VolSDF", seed=0, beta=0.01, device="cuda", precision="bf16x3")
blob, _ = model.packed()
pts = (torch.rand(3 * (1 << 20), 3, generator=torch.Generator().manual_seed(0)) * 4 - 2).cuda()
for want in (True, False):
    hip.sdf_nabla_fwd(blob, pts, 3.0, want_h7=want, precision=1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): hip.sdf_nabla_fwd(blob, pts, 3.0, want_h7=want, precision=1)
    e1.record(); torch.cuda.synchronize()
    print("MS", "h7" if want else "no_h7", e0.elapsed_time(e1) / 5)
''' % ROOT
    res = {}
    for name, lib in [("full", None)] + [(n, os.path.join(OUT, f"libgrad_{n}.so")) for n in VARIANTS]:
        env = dict(os.environ)
        if lib:
            env["NERFART_HIP_LIB"] = lib
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        for l in r.stdout.splitlines():
            if l.startswith("MS"):
                res[f"{name}_{l.split()[1]}"] = round(float(l.split()[2]), 3)
        if not r.stdout.strip():
            res[name] = r.stderr[-300:]
    print(json.dumps({"ms_per_3M_points": res}))


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
