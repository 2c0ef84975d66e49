"""Build libnerfart_hip.so (the C-ABI library of hand-written gfx950 kernels) in-tree with hipcc.

    python -m nerfart_amd.build          # or: from nerfart_amd.build import build; build()

Sources live in nerfart_amd/csrc; objects go to csrc/_build, the library to
csrc/libnerfart_hip.so (git-ignored, travels to the GPU box with the snapshot).
hipcc cross-compiles for gfx950 without a GPU present.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libnerfart_hip.so")
SOURCES = ["capi_common.cpp", "wgrad.hip", "pass2_operands.hip", "render_backward.hip", "mlp_chain.hip", "mlp_chain_bf16.hip", "mlp_chain_f16x2.hip", "mlp_chain_f16x1.hip", "mlp_k2_w32.hip", "mlp_grad_bf16.hip", "mlp_backward_bf16.hip", "volsdf_render.hip", "volsdf_backward.hip", "neus_render.hip", "raygen.hip", "clip_vit.hip", "style_heads.hip", "ray_casting.hip", "vgg_conv.hip", "pack_blob.hip", "geo_feature.hip"]
HEADERS = ["nerfart_common.h", "ray_common.h", "mlp_common.h", "mlp_bf16_core.h", "gemm_f16.h", "gemm_f32.h"]
INCLUDES = {"mlp_chain_f16x2.hip": ["mlp_chain_bf16.hip", "mlp_grad_bf16.hip"], "mlp_chain_f16x1.hip": ["mlp_chain_bf16.hip"]}      # sources compiled a second time (precision 4)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-x", "hip"]
# TEST-ONLY variant libraries (never loaded by the product: nerfart_amd.hip binds libnerfart_hip.so): the same objects with ONE source compiled with
# extra defines.  scan_generic: the per-ray sampler kernels with the generic two-pass error-bound scan instead of the register-cached one -
# tests/test_gpu_guarded_sampler.py holds the two bit-identical on whole frames (csrc/ray_common.h).
VARIANTS = {"scan_generic": ("volsdf_render.hip", ["-DNERFART_SCAN_GENERIC"])}


def variant_lib(name: str) -> str:
    return os.path.join(CSRC, f"libnerfart_hip_{name}.so")


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose: bool = True, force: bool = False) -> str:
    hipcc = _hipcc()
    bdir = os.path.join(CSRC, "_build")
    os.makedirs(bdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(bdir, os.path.splitext(s)[0] + ".o")
        deps = [src] + hdrs + [os.path.join(CSRC, i) for i in INCLUDES.get(s, [])]
        if force or _stale(obj, deps):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print("[nerfart build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[nerfart build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    for name, (src_name, defs) in VARIANTS.items():
        src = os.path.join(CSRC, src_name)
        obj = os.path.join(bdir, os.path.splitext(src_name)[0] + f".{name}.o")
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + FLAGS + defs + ["-c", src, "-o", obj]
            if verbose:
                print("[nerfart build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        vlib = variant_lib(name)
        others = [o for o in objs if os.path.basename(o) != os.path.splitext(src_name)[0] + ".o"]
        if force or _stale(vlib, others + [obj]):
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", vlib] + others + [obj]
            if verbose:
                print("[nerfart build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
