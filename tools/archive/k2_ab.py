#!/usr/bin/env python
"""A/B of the two K2 (SDF-only, split-bf16) kernels on one MI355X: the one-wave-per-SIMD kernel (NERFART_K2=w32, csrc/mlp_k2_w32.hip)
against the 8-wave kernel (default, csrc/mlp_chain_bf16.hip).  Each variant runs in its own process (the switch is read once);
prints ms per 4 M-point launch, algorithmic TFLOP/s and whether the two outputs are bit-identical.
    python tools/archive/k2_ab.py [--points 4194304] [--reps 10]"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(args):
    import torch
    from nerfart_amd import scene, hip
    model, _, _ = scene.build_model("VolSDF", seed=0, beta=0.01, device="cuda", precision="bf16x3")
    blob, _ = model.packed()
    g = torch.Generator().manual_seed(0)
    pts = (torch.rand(args.points, 3, generator=g) * 4 - 2).cuda()
    out = hip.sdf_fwd(blob, pts, 3.0, precision=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        out = hip.sdf_fwd(blob, pts, 3.0, precision=1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    torch.save(out.cpu(), args.child)
    # ragged sizes too (tile tails)
    small = [hip.sdf_fwd(blob, pts[:n].contiguous(), 3.0, precision=1).cpu() for n in (1, 17, 127, 129, 1000)]
    torch.save(small, args.child + ".small")
    print(json.dumps({"variant": os.environ.get("NERFART_K2", "v1"), "ms": round(ms, 4), "tflops_algorithmic": round(args.points * 1049088 / ms / 1e9, 1)}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1 << 22)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--child", default=None)
    args = ap.parse_args()
    if args.child:
        return child(args)
    import torch
    outs = {}
    for var in ("w32", "v1"):
        env = dict(os.environ)
        env["NERFART_K2"] = var
        path = f"/tmp/k2_{var}.pt"
        r = subprocess.run([sys.executable, __file__, "--points", str(args.points), "--reps", str(args.reps), "--child", path], env=env,
                           capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-2000:])
        outs[var] = (torch.load(path), torch.load(path + ".small"))
    same = torch.equal(outs["w32"][0], outs["v1"][0]) and all(torch.equal(a, b) for a, b in zip(outs["w32"][1], outs["v1"][1]))
    d = (outs["w32"][0] - outs["v1"][0]).abs().max().item()
    print(json.dumps({"bit_identical": same, "max_abs_diff": d}))


if __name__ == "__main__":
    main()
