#!/usr/bin/env python
"""Is the sdf error of a cheaper sampler arithmetic NOISE (different at neighbouring samples of a ray: the CDF averages it) or a COHERENT shift (the same
for neighbours: every fine sample of the ray moves together)?  CPU model of the kernels' arithmetic (tools/emul_sampler_precision.py) on 128 rays x 4,096
samples (depth spacing 1e-3) of orbit pose 1: rms of the pointwise error near the surface (|sdf| < 0.05), and rms / max of its mean over windows of 16
consecutive samples (noise shrinks 4x, a shift does not).  lo_layers: the hidden layers whose weights keep their lo term.

    python tools/coherent_error.py > profiles/r09_coherent_error.txt
"""
import sys, os, torch, numpy as np, torch.nn.functional as F
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests/golden'); sys.path.insert(0, '/root/repo/tools')
import make_oracle_views as mov
from oracle import nets
sd = mov.scene_sd()
def q(x, dt=torch.float16): return x.to(dt).float()
def surface_v(sd, x, lo_layers, act_q=True, multires=6, skips=(4,), stem="implicit_surface.surface_fc_layers"):
    e = nets.embed(x, multires); h = e
    D = nets.n_layers(sd, stem) - 1
    for i in range(D):
        W, b = nets.folded_weight(sd, f"{stem}.{i}"), sd[f"{stem}.{i}.bias"]
        if i == 0: z = F.linear(e, W, b)
        else:
            if i in skips:
                W = W / np.sqrt(2); nh = h.shape[-1]; Wh, We = W[:, :nh], W[:, nh:]
            else: Wh, We = W, None
            Whq = q(Wh) + (q(Wh - q(Wh)) if i in lo_layers else 0)
            z = F.linear(q(h) if act_q else h, Whq, b)
            if We is not None: z = z + F.linear(e, We)
        h = nets.softplus100(z)
    W, b = nets.folded_weight(sd, f"{stem}.{D}"), sd[f"{stem}.{D}.bias"]
    return F.linear(h, W[:1], b[:1])[..., 0]
idx = torch.arange(0, mov.N, mov.N // 128)[:128]
_, o, d = mov.view_rays(1, idx)
d = F.normalize(d, dim=-1)
t = torch.linspace(0.5, 4.5, 4096)
pts = (o[:, None, :] + d[:, None, :] * t[None, :, None]).reshape(-1, 3)
with torch.no_grad():
    exact = nets.surface_forward(sd, pts)[0].reshape(128, 4096)
    near = exact.abs() < 0.05
    print("near-surface samples:", int(near.sum()))
    allL = set(range(1, 8))
    for name, lo, aq in (("x2", allL, True), ("x1", set(), True), ("weights-only x1 (exact act)", set(), False), ("act-only (exact w)", allL, True),
                         ("lo on 7", {7}, True), ("lo on 6,7", {6, 7}, True), ("lo on 5,6,7", {5, 6, 7}, True), ("lo on 4..7", {4,5,6,7}, True), ("lo on 1,2,3", {1,2,3}, True), ("lo on 1", {1}, True), ("lo on 1,2", {1,2}, True)):
        v = surface_v(sd, pts, lo, aq).reshape(128, 4096)
        err = v - exact
        win = F.avg_pool1d(err[:, None, :], 16, 16)[:, 0]          # mean over 16 consecutive samples (1.6e-2 of depth)
        nw = F.avg_pool1d(near.float()[:, None, :], 16, 16)[:, 0] > 0.5
        print(f"{name:28s} pointwise rms {float(err[near].pow(2).mean().sqrt()):.2e} | window-mean (coherent) rms {float(win[nw].pow(2).mean().sqrt()):.2e} max {float(win[nw].abs().max()):.2e}")
