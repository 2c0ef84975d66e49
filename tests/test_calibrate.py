"""nerfart_amd/calibrate.py on the CPU: the error-compensated one-term fp16 rounding (second session of round 6).  The calibration points' near-surface
half is selected with the HIP SDF kernel in the product; here the oracle's SDF stands in (no GPU), everything else is the module's own code."""
import numpy as np
import torch


def test_compensated_rounding_cancels_the_dropped_product_and_stays_on_the_fp16_grid():
    from nerfart_amd import calibrate, frameworks, scene
    from oracle import nets
    torch.manual_seed(0)
    model, _, _, _, _ = frameworks.get_model(scene.synthetic_config("VolSDF"))
    model.load_state_dict(scene.perturb_state(model.state_dict(), beta=0.01, seed=1))
    sd = {k: v.detach() for k, v in model.state_dict().items()}

    def points(model_, n=4096, seed=0):
        g = torch.Generator().manual_seed(seed)
        u = torch.randn(16 * n, 3, generator=g)
        u = u / u.norm(dim=-1, keepdim=True) * (torch.rand(16 * n, 1, generator=g) ** (1 / 3)) * 3.0
        with torch.no_grad():
            s = nets.surface_forward(sd, u)[0]
        near = u[s.abs() < calibrate.NEAR][: n // 2]
        return torch.cat([u[: n - near.shape[0]], near])
    saved = calibrate.calibration_points
    calibrate.calibration_points = points
    try:
        g, v, b, stats = calibrate.compensated_surface_layers(model, n=4096)
    finally:
        calibrate.calibration_points = saved
    assert sorted(stats) == [1, 2, 3, 4, 5, 6, 7]
    for i, (nearest, comp) in stats.items():
        assert comp < 0.35 * nearest, (i, nearest, comp)           # measured 0.12 - 0.21 at 12,288 points
    c = calibrate.C_SCALE                 # the kernel's scaled softplus recursion: z' = c z, a' = c softplus(z)
    for i in range(9):
        W = nets.folded_weight(sd, f"implicit_surface.surface_fc_layers.{i}")
        fold = g[i].reshape(-1, 1) * v[i] / v[i].norm(dim=1, keepdim=True)       # what nerfart_pack_surface_blob computes from (weight_g, weight_v)
        bias = sd[f"implicit_surface.surface_fc_layers.{i}.bias"]
        assert torch.allclose(b[i], bias * (c if i < 8 else 1.0), rtol=1e-6, atol=1e-9), "hidden biases carry c, the last layer's does not"
        if i in (0, 8):
            assert torch.allclose(fold, W * (c if i == 0 else 1.0 / c), rtol=1e-5, atol=1e-9), "layer 0 carries c (its inputs are the raw encoding), the sdf row 1 / c; no rounding"
            continue
        nh = 217 if i == 4 else W.shape[1]
        sc = np.float32(1 / np.sqrt(2)) if i == 4 else np.float32(1.0)             # the packer folds the skip layer's 1 / sqrt 2 into its weights
        hid = fold[:, :nh] * sc
        assert float((hid.half().float() - hid).abs().max() / hid.abs().max()) < 1e-6, "hidden columns sit on the fp16 grid: their hi fragments hold them exactly"
        w0 = W[:, :nh] * sc
        # a compensated weight absorbs its predecessors' residuals: it moves by a few fp16 steps OF THE LAYER'S LARGE WEIGHTS at most (small weights
        # drift by many of their own, finer steps - the correction is absolute), never further
        assert float((hid - w0).abs().max()) < 2.0 ** -9 * float(w0.abs().max())
        moved = float((hid.half() != (W[:, :nh] * sc).half()).float().mean())
        assert 0.05 < moved < 0.5, moved
        if i == 4:
            assert torch.allclose(fold[:, nh:], W[:, nh:] * c, rtol=1e-5), "the skip layer's encoding columns keep their hi + lo weights (times c: they multiply the raw encoding)"
    # the scaled recursion is the same network: z'_l = c z_l, a'_l = c a_l, sdf unchanged (fp64 restatement with the returned tensors)
    x = torch.rand(256, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(2)) * 2 - 1
    e = calibrate._embed(x, 6)
    h = e
    for i in range(8):
        Wi = (g[i].reshape(-1, 1) * v[i] / v[i].norm(dim=1, keepdim=True)).double()
        if i == 4:
            Wi = Wi / np.sqrt(2.0)
            h = torch.cat([h, e], dim=-1)
        z = h @ Wi.T + b[i].double()
        h = torch.clamp(z, min=0) + torch.log2(1 + torch.exp2(-z.abs()))
    sdf = h @ (g[8].reshape(-1, 1) * v[8] / v[8].norm(dim=1, keepdim=True)).double()[0] + b[8].double()[0]
    ref = nets.surface_forward({k: t.double() for k, t in sd.items()}, x)[0]
    assert float((sdf - ref).abs().max()) < 2e-3, float((sdf - ref).abs().max())       # one-term weights: 1e-4-class, not bits


def test_compensated_round_fp16_is_nearest_rounding_when_nothing_correlates():
    """White inputs (H = I): no column can absorb another's residual - the sequential rule degenerates to round-to-nearest."""
    from nerfart_amd import calibrate
    g = torch.Generator().manual_seed(1)
    W = torch.randn(8, 32, generator=g, dtype=torch.float64) * 0.1
    X = torch.eye(32, dtype=torch.float64).repeat(4, 1)
    Q = calibrate.compensated_round_fp16(W, X, damp=0.0)
    assert torch.equal(Q, W.float().half().double())
