"""Pack-time calibration of ONE-TERM fp16 weights for Algorithm 1's sampler (C-ABI precision 5, csrc/mlp_chain_f16x1.hip): error-compensated rounding.

Why.  The 1-MFMA kernel multiplies one fp16 activation term with one fp16 weight term.  Rounding every weight to the NEAREST fp16 value drops the
product (W - fp16 W) . a of every hidden layer, and because a rounded weight is the same for every point, what is dropped is a COHERENT shift of the
SDF along a ray (tools/coherent_error.py: 6.0e-5 rms over windows of 16 consecutive samples against 2.2e-5 with hi + lo weights) - the one error the
sampler's guard does not catch.  But the sum that is dropped is over 256 inputs whose activations are strongly correlated over the
scene, and each weight has TWO neighbouring fp16 values to land on: choosing them so that the dropped products cancel over the scene's own
activations (the sequential, second-order rounding of OBQ / GPTQ: round column k to nearest, push its residual onto the not-yet-rounded columns through
the inverse of the activations' Gram matrix H = A A^T) takes the dropped product from 3.5e-5 to 5e-6 rms per layer and the coherent sdf error to
2.5e-5 - the 2-MFMA kernel's, i.e. the 11-bit activations' own floor.

What runs where.  This is a property of a set of WEIGHTS, computed once per set (about a second; `nets._PackedModel.packed_sampler` caches it by
parameter version) - never per frame, never in training (the trainer's sampler blob changes every step and stays hi + lo).  Calibration points are drawn
in the model's bounding sphere and, half of them, near its surface - selected with the HIP SDF kernel; the layer inputs at those points and the seven
256 x 256 solves are plain fp64 linear algebra on the host CPU (torch as a LAPACK front end; no GPU library call, no part of any render or training path).
The result goes through the ordinary C-ABI packer (nerfart_pack_surface_blob, precision 5: the fp16 hi + lo layout under its own encoding word) as folded
weights that already sit on the fp16 grid: its hi fragments of the hidden k-steps hold them exactly, the encodings' k-steps keep hi + lo.

The tensors are handed over in the kernel's SCALED recursion (C_SCALE = 100 log2 e; mlp_bf16_core.h, NERFART_F16X1): accumulators hold z' = c z, activations
a' = c softplus(z) = max(z', 0) + log2(1 + 2^-|z'|) - one multiply per value less in an epilogue that costs the 1-MFMA kernel more than its matrix work.  Layer 0's
weights, the skip layer's encoding columns and the hidden biases carry c, the sdf row 1 / c, the hidden weights are the network's own (they multiply c a).
"""
from __future__ import annotations

import math

import torch

from . import hip

C_SCALE = 100.0 * math.log2(math.e)      # the 1-MFMA kernel's scaled softplus recursion (mlp_bf16_core.h, NERFART_F16X1): accumulators hold z' = c z, activations a' = c a
N_CALIB = 12288          # calibration points: half uniform in the bounding sphere, half with |sdf| < NEAR
NEAR = 0.1
DAMP = 0.01              # GPTQ damping: lambda = DAMP * mean(diag H)


def _fold(weight_g: torch.Tensor, weight_v: torch.Tensor) -> torch.Tensor:
    """weight_norm: g * v / ||v||_row (nn.utils.weight_norm, dim 0) in fp64 on the host."""
    v = weight_v.double()
    return weight_g.double().reshape(-1, 1) * v / v.norm(dim=1, keepdim=True)


def _embed(x: torch.Tensor, multires: int) -> torch.Tensor:
    """reference Embedder (models/base.py:38-64): [x, sin(2^k x), cos(2^k x)]_k"""
    out = [x]
    for k in range(multires):
        out += [torch.sin(x * (2.0 ** k)), torch.cos(x * (2.0 ** k))]
    return torch.cat(out, dim=-1)


def _softplus100(z: torch.Tensor) -> torch.Tensor:
    return torch.clamp(z, min=0) + torch.log1p(torch.exp(-100.0 * z.abs())) / 100.0


def compensated_round_fp16(W: torch.Tensor, X: torch.Tensor, damp: float = DAMP) -> torch.Tensor:
    """W [out, K] fp64 -> Wq [out, K] on the fp16 grid, minimising ||(W - Wq) X^T||_F over the calibration inputs X [N, K] (fp64): columns in order,
    each rounded to nearest and its residual propagated to the later columns through the Cholesky factor of H^-1, H = X^T X / N + damping."""
    W = W.clone()
    K = W.shape[1]
    H = X.T @ X / X.shape[0]
    H = H + damp * H.diag().mean() * torch.eye(K, dtype=H.dtype)
    U = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H)), upper=True)
    Q = torch.empty_like(W)
    for k in range(K):
        w = W[:, k]
        q = w.float().half().double()
        Q[:, k] = q
        if k + 1 < K:
            W[:, k + 1:] -= ((w - q) / U[k, k])[:, None] * U[k, k + 1:][None, :]
    return Q


def calibration_points(model, n: int = N_CALIB, seed: int = 0) -> torch.Tensor:
    """[n, 3] points (CPU, fp32): half uniform in the model's bounding sphere, half near its surface (|sdf| < NEAR, found with the HIP SDF kernel)."""
    R = float(getattr(model, "obj_bounding_radius", 1.0))
    dev = next(model.parameters()).device
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(16 * n, 3, generator=g)
    u = u / u.norm(dim=-1, keepdim=True) * (torch.rand(16 * n, 1, generator=g) ** (1.0 / 3.0)) * R
    blob, _ = model.packed()
    with torch.no_grad():
        sdf = hip.sdf_fwd(blob, u.to(dev).contiguous(), R, precision=model.precision_id).cpu()
    near = u[sdf.abs() < NEAR][: n // 2]
    return torch.cat([u[: n - near.shape[0]], near])


def compensated_surface_layers(model, n: int = N_CALIB, seed: int = 0, compensate: bool = True):
    """(weight_g, weight_v, bias) lists for hip.pack_surface_blob(5, ...) - the blob of the 1-MFMA K2 (C-ABI precision 5): per layer the FOLDED weight matrix
    as `weight_v` with its row norms as `weight_g` (the packer's fold g v / ||v|| then reproduces it to an fp32 ulp, far inside an fp16 step), in the SCALED
    recursion the kernel runs (z' = c z, a' = c softplus(z), c = 100 log2 e: layer 0's weights, the skip layer's encoding columns and every hidden bias times
    c, the last layer's row divided by c, the hidden layers' own weights unchanged), hidden-layer columns on the fp16 grid -
    compensate=True: error-compensated against the model's activations (what model.calibrate_sampler() ships); False: rounded to nearest (the measured,
    not shipped form of DESIGN.md 4.1e; no calibration points are drawn).
    Also returns a dict of per-layer statistics (rms of the dropped product on the calibration set: nearest / compensated)."""
    S = model.implicit_surface
    L = list(S.surface_fc_layers)
    dev = L[0].weight_v.device
    D, skips, multires = S.D, tuple(S.skips), S.embed_multires
    c = C_SCALE
    # at most 8 host threads (what the ~2 s were measured on): one process per GPU calibrates its own copy, concurrently with the other ranks'
    n_threads = torch.get_num_threads()
    torch.set_num_threads(min(n_threads, 8))
    try:
        return _compensated_surface_layers(model, L, dev, D, skips, multires, c, n, seed, compensate)
    finally:
        torch.set_num_threads(n_threads)


def _compensated_surface_layers(model, L, dev, D, skips, multires, c, n, seed, compensate):
    with torch.no_grad():
        Wf = [_fold(l.weight_g.detach().cpu(), l.weight_v.detach().cpu()) for l in L]
        bf = [l.bias.detach().cpu().double() for l in L]
        out_w, out_b, stats = [], [], {}
        if compensate:
            pts = calibration_points(model, n, seed).double()
            e = _embed(pts, multires)
            h = e
        for i in range(D):
            W = Wf[i]
            out_b.append(bf[i] * c)
            if i == 0:
                out_w.append(W * c)                                     # ready-made input units (hi + lo in the kernel): z'_0 = (c W_0) enc + c b_0
                if compensate:
                    h = _softplus100(h @ W.T + bf[i])
                continue
            nh = Wf[i - 1].shape[0]                                     # the previous layer's outputs = this layer's hidden inputs
            scale = 1.0 / math.sqrt(2.0) if i in skips else 1.0       # cat[h, enc] / sqrt 2 (base.py:248-250): the packer folds it into layer i's weights
            Wh = W[:, :nh] * scale
            if compensate:
                hq = (h * c).float().half().double()                    # the kernel's B operand: one fp16 term of the scaled activation a' = c a
                Wq = compensated_round_fp16(Wh, hq)
                stats[i] = (float(((Wh - Wh.float().half().double()) @ hq.T).pow(2).mean().sqrt()) / c, float(((Wh - Wq) @ hq.T).pow(2).mean().sqrt()) / c)
            else:
                Wq = Wh.float().half().double()
            # the skip layer's encoding columns multiply the raw encoding: they carry c (and keep hi + lo); the packer applies 1 / sqrt 2 to the whole tensor
            out_w.append(torch.cat([Wq / scale, W[:, nh:] * c], dim=1) if i in skips else Wq)
            if compensate:
                x = torch.cat([h, e], dim=-1) * scale if i in skips else h
                h = _softplus100(x @ W.T + bf[i])
        out_w.append(Wf[D] / c)                                         # sdf (and the unused feature rows) = (W_8 / c) a'_7 + b_8
        out_b.append(bf[D])
    g = [w.norm(dim=1).float().reshape(-1, 1).to(dev) for w in out_w]
    v = [w.float().contiguous().to(dev) for w in out_w]
    b = [x.float().contiguous().to(dev) for x in out_b]
    return g, v, b, stats
