#!/usr/bin/env python
"""Timing ablation of k_sdf_only_bf16: builds variant libraries with one component compiled out
(weight DMA / epilogue / MFMA) and times nerfart_sdf_fwd on 4M points with each.  Results are WRONG by
construction - this only attributes time.   build:  python tools/ablate_bf16.py build ;  run (GPU): ... run"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nerfart_amd", "csrc")
VARIANTS = {"full": [], "nodma": ["-DNERFART_ABLATE_DMA"], "noepi": ["-DNERFART_ABLATE_EPI"], "nomfma": ["-DNERFART_ABLATE_MFMA"],
            "nodma_noepi": ["-DNERFART_ABLATE_DMA", "-DNERFART_ABLATE_EPI"], "noldsread": ["-DNERFART_ABLATE_LDSREAD"],
            "mfma_only": ["-DNERFART_ABLATE_DMA", "-DNERFART_ABLATE_EPI", "-DNERFART_ABLATE_LDSREAD"],
            "nobarrier": ["-DNERFART_ABLATE_BARRIER"], "nobarrier_novmwait": ["-DNERFART_ABLATE_BARRIER", "-DNERFART_ABLATE_VMWAIT"],
            "mfma_only_nobarrier": ["-DNERFART_ABLATE_DMA", "-DNERFART_ABLATE_EPI", "-DNERFART_ABLATE_LDSREAD", "-DNERFART_ABLATE_BARRIER"]}
if os.environ.get("NERFART_ABLATE_SET") == "exp":      # scheduling experiments (results stay correct except where noted)
    VARIANTS = {"full": [], "prio_young": ["-DNERFART_EXP_PRIO_YOUNG"], "prio_old": ["-DNERFART_EXP_PRIO_OLD"],
                "nosched": ["-DNERFART_EXP_NOSCHED"]}
if os.environ.get("NERFART_ABLATE_SET") == "pair":     # tiles multiplied in pairs, accumulator chains interleaved (results correct)
    VARIANTS = {"full": [], "pair": ["-DNERFART_EXP_PAIR"], "pair_mfma_only": ["-DNERFART_EXP_PAIR", "-DNERFART_ABLATE_DMA", "-DNERFART_ABLATE_EPI", "-DNERFART_ABLATE_LDSREAD"]}
if os.environ.get("NERFART_ABLATE_SET") == "dma":      # where the LDS-DMA pieces sit among the items (k_sdf_only: results correct)
    VARIANTS = {"full": [], "t0_0": ["-DNERFART_EXP_DMA_T0=0"], "t0_2": ["-DNERFART_EXP_DMA_T0=2"], "t0_12": ["-DNERFART_EXP_DMA_T0=12"],
                "spread0": ["-DNERFART_EXP_DMA_SPREAD=0"], "spread2": ["-DNERFART_EXP_DMA_SPREAD=2"]}
if os.environ.get("NERFART_ABLATE_SET") == "epi2":     # two independent epilogue pairs per slice (results correct)
    VARIANTS = {"full": [], "epi2": ["-DNERFART_EXP_EPI2"]}
if os.environ.get("NERFART_ABLATE_SET") == "iso":      # the non-MFMA stream in isolation, and a non-temporal weight stream
    VARIANTS = {"full": [], "dma_nt": ["-DNERFART_EXP_DMA_NT"],
                "nomfma_nodma": ["-DNERFART_ABLATE_MFMA", "-DNERFART_ABLATE_DMA"], "nomfma_noepi": ["-DNERFART_ABLATE_MFMA", "-DNERFART_ABLATE_EPI"],
                "nomfma_noldsread": ["-DNERFART_ABLATE_MFMA", "-DNERFART_ABLATE_LDSREAD"], "dma_only": ["-DNERFART_ABLATE_MFMA", "-DNERFART_ABLATE_EPI", "-DNERFART_ABLATE_LDSREAD"]}
if os.environ.get("NERFART_ABLATE_SET") == "skel":     # what is left when every per-item component is compiled out
    A = ["-DNERFART_ABLATE_MFMA", "-DNERFART_ABLATE_EPI", "-DNERFART_ABLATE_LDSREAD"]
    VARIANTS = {"dma_only": A, "skeleton": A + ["-DNERFART_ABLATE_DMA"], "skeleton_nobarrier": A + ["-DNERFART_ABLATE_DMA", "-DNERFART_ABLATE_BARRIER"],
                "skeleton_nobarrier_novmwait": A + ["-DNERFART_ABLATE_DMA", "-DNERFART_ABLATE_BARRIER", "-DNERFART_ABLATE_VMWAIT"]}
if os.environ.get("NERFART_ABLATE_SET") == "seg4":     # segments of four tiles: 12 independent MFMAs, then the fillers (results correct)
    VARIANTS = {"full": [], "seg4": ["-DNERFART_EXP_SEG4"]}
if os.environ.get("NERFART_ABLATE_SET") == "ahead":    # fragment prefetch distance (results correct)
    VARIANTS = {"full": [], "ahead3": ["-DNERFART_AHEAD=3"]}
if os.environ.get("NERFART_ABLATE_SET") == "agpr":     # accumulators in AGPRs (results correct)
    VARIANTS = {"full": [], "agpr": ["-DNERFART_EXP_AGPR"]}
if os.environ.get("NERFART_ABLATE_SET") == "one":      # just the current sources (compare with a previous run's "full")
    VARIANTS = {"full": []}
if os.environ.get("NERFART_ABLATE_SET") == "early":    # all DMA pieces of a chunk in its first k-step (results correct)
    VARIANTS = {"full": [], "dma_early": ["-DNERFART_EXP_DMA_EARLY"]}
OUT = os.path.join(ROOT, "gpurun_ablate")

def build():
    os.makedirs(OUT, exist_ok=True)
    srcs = ["capi_common.cpp", "mlp_chain.hip", "mlp_chain_bf16.hip", "mlp_grad_bf16.hip", "mlp_backward_bf16.hip", "volsdf_render.hip", "volsdf_backward.hip", "neus_render.hip", "raygen.hip"]
    for name, flags in VARIANTS.items():
        lib = os.path.join(OUT, f"lib_{name}.so")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-shared", "-x", "hip"] + flags + \
              [os.path.join(CSRC, s) for s in srcs] + ["-o", lib]
        print(" ".join(cmd[-3:]), flush=True)
        subprocess.check_call(cmd)

def run():
    import json
    res = {}
    for name in VARIANTS:
        env = dict(os.environ, NERFART_HIP_LIB=os.path.join(OUT, f"lib_{name}.so"))
        code = r'''
import sys, time, torch
sys.path.insert(0, %r)
from nerfart_amd import scene, hip
m, rk, fn = scene.build_model("VolSDF", device="cuda", precision="bf16x3")
blob, _ = m.packed()
pts = torch.rand(4*1024*1024, 3, device="cuda") * 4 - 2
for _ in range(2): hip.sdf_fwd(blob, pts, 3.0, precision=1)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): hip.sdf_fwd(blob, pts, 3.0, precision=1)
torch.cuda.synchronize(); print((time.perf_counter() - t) / 5 * 1e3)
''' % ROOT
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        try:
            res[name] = float(out.stdout.strip().splitlines()[-1])
        except Exception:
            res[name] = out.stderr[-300:]
        print(name, res[name], flush=True)
    print(json.dumps(res))

if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
