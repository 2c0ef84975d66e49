#!/usr/bin/env python
"""CPU probe that preceded nerfart_amd/calibrate.py (second session of round 6): sequential error-compensated rounding (GPTQ / OBQ) of the one-term fp16 weights against the
activations of 12 k calibration points, and what it does to the COHERENT sdf error along test rays (tools/coherent_error.py) - profiles/r10_compensated_rounding_probe.txt."""
import sys, os, torch, numpy as np, torch.nn.functional as F
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests/golden')
import make_oracle_views as mov
from oracle import nets
torch.manual_seed(0)
import os
def _sd(seed):
    from nerfart_amd import scene, frameworks
    torch.manual_seed(seed)
    model, _, _, _, _ = frameworks.get_model(scene.synthetic_config("VolSDF"))
    return scene.perturb_state(model.state_dict(), beta=0.01, seed=seed + 1)
sd = _sd(int(os.environ.get("SCENE_SEED", "0")))
stem = "implicit_surface.surface_fc_layers"
def q(x, dt=torch.float16): return x.to(dt).float()

def gptq_round(W, X, damp=0.01):
    """W [out, K] fp32, X [N, K] calibration inputs (already fp16-rounded activations).  Returns Wq on the fp16 grid minimising ||(W - Wq) X^T||."""
    W = W.double().clone(); K = W.shape[1]
    H = (X.double().T @ X.double()) / X.shape[0]
    H += damp * H.diag().mean() * torch.eye(K, dtype=torch.float64)
    Hinv = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H)), upper=True)
    Q = torch.zeros_like(W)
    for i in range(K):
        w = W[:, i]
        qi = w.float().half().double()
        Q[:, i] = qi
        err = (w - qi) / Hinv[i, i]
        W[:, i + 1:] -= err[:, None] * Hinv[i, i + 1:][None, :]
    return Q.float()

def forward(x, mode, Wq=None, collect=None):
    e = nets.embed(x, 6); h = e
    D = nets.n_layers(sd, stem) - 1
    for i in range(D):
        W, b = nets.folded_weight(sd, f"{stem}.{i}"), sd[f"{stem}.{i}.bias"]
        if i == 0: z = F.linear(e, W, b)
        else:
            if i == 4:
                W = W / np.sqrt(2); nh = h.shape[-1]; Wh, We = W[:, :nh], W[:, nh:]
            else: Wh, We = W, None
            hq = q(h)
            if collect is not None: collect[i] = (Wh, hq)
            if mode == "x2": Whq = q(Wh) + q(Wh - q(Wh))
            elif mode == "x1": Whq = q(Wh)
            else: Whq = Wq[i]
            z = F.linear(hq, Whq, b)
            if We is not None: z = z + F.linear(e, We)
        h = nets.softplus100(z)
    W, b = nets.folded_weight(sd, f"{stem}.{D}"), sd[f"{stem}.{D}.bias"]
    return F.linear(h, W[:1], b[:1])[..., 0]

# calibration points: uniform in the ball R = 3 and near-surface points
g = torch.Generator().manual_seed(7)
u = torch.randn(1 << 17, 3, generator=g); u = u / u.norm(dim=-1, keepdim=True) * (torch.rand(1 << 17, 1, generator=g) ** (1 / 3)) * 3.0
with torch.no_grad():
    su = nets.surface_forward(sd, u)[0]
near = u[su.abs() < 0.1]
calib = torch.cat([u[:8192], near[:8192]])
print("calibration:", calib.shape[0], "points,", min(8192, near.shape[0]), "near the surface")
# sequential calibration: layer by layer on the quantised network so far
Wq = {}
with torch.no_grad():
    for i in range(1, 8):
        col = {}
        forward(calib, "gptq_partial", {**{k: v for k, v in Wq.items()}, **{j: q(nets.folded_weight(sd, f"{stem}.{j}")[:, :217] / np.sqrt(2)) if j == 4 else q(nets.folded_weight(sd, f"{stem}.{j}")) for j in range(i, 8)}}, collect=col)
        Wh, hq = col[i]
        Wq[i] = gptq_round(Wh, hq)
        d0 = ((Wh - q(Wh)) @ hq.T).pow(2).mean().sqrt(); d1 = ((Wh - Wq[i]) @ hq.T).pow(2).mean().sqrt()
        print(f"layer {i}: rms of the dropped product on the calibration set: nearest {float(d0):.2e} -> compensated {float(d1):.2e}; weights moved off nearest: {float((Wq[i] != q(Wh)).float().mean()):.2f}")
idx = torch.arange(0, mov.N, mov.N // 128)[:128]
for pose in (1, 23):
    _, o, d = mov.view_rays(pose, idx); d = F.normalize(d, dim=-1)
    t = torch.linspace(0.5, 4.5, 4096)
    pts = (o[:, None, :] + d[:, None, :] * t[None, :, None]).reshape(-1, 3)
    with torch.no_grad():
        exact = nets.surface_forward(sd, pts)[0].reshape(128, 4096)
        nearm = exact.abs() < 0.05
        for name in ("x2", "x1", "gptq"):
            v = forward(pts, name, Wq).reshape(128, 4096)
            err = v - exact
            win = F.avg_pool1d(err[:, None, :], 16, 16)[:, 0]
            nw = F.avg_pool1d(nearm.float()[:, None, :], 16, 16)[:, 0] > 0.5
            print(f"pose {pose} {name:5s} pointwise rms {float(err[nearm].pow(2).mean().sqrt()):.2e} | window-mean (coherent) rms {float(win[nw].pow(2).mean().sqrt()):.2e} max {float(win[nw].abs().max()):.2e}")

