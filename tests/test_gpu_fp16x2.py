"""C-ABI precision 4 ("fp16x2", csrc/mlp_chain_f16x2.hip): the 2-MFMA measurement variant of the three forward kernels.  Held to its own
CPU emulation (tests/emul_chain.py with TERM = "fp16" walking the real fp16 blob: same data flow, so agreement is at fp32 summation
noise) and, loosely, to the oracle; the pixel statistics that decide whether it could ever ship are tools/parity_table.py's."""
import numpy as np
import pytest
import torch

from conftest import scene_state

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_fp16x2_kernels_match_their_emulation_and_the_oracle():
    import emul_chain as em
    from nerfart_amd import hip, packing, scene
    from oracle import nets
    model, _, _ = scene.build_model("VolSDF", seed=0, beta=0.01, device=DEV, precision="fp16x2")
    assert model.precision_id == 4
    surf, rad = model.packed()
    sd, _ = scene_state("VolSDF", 0.01)
    surf_cpu = packing.surface_plan_bf16(term="fp16").pack(packing.surface_tensors(sd)).numpy()
    # the GPU folds weight_norm with its own rounding (1 fp32 ulp): a hi term may land on the neighbouring fp16 value, its lo term follows
    hdr = surf_cpu[:512].view(np.int32)
    wa = surf.cpu().numpy()[512:int(hdr[4])].view(np.float16).astype(np.float32).reshape(-1, 2, 64, 8)
    wb = surf_cpu[512:int(hdr[4])].view(np.float16).astype(np.float32).reshape(-1, 2, 64, 8)
    assert np.abs(wa.sum(1) - wb.sum(1)).max() < 1e-6, "GPU- and CPU-packed fragments (hi + lo) are the same weights"
    g = torch.Generator().manual_seed(23)
    pts = (torch.rand(64, 3, generator=g) * 4 - 2)
    pts[:4] *= 2.0
    view = torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=-1)
    x = pts.to(DEV)
    s_k2 = hip.sdf_fwd(surf, x, 3.0, precision=4).cpu().numpy()
    sdf, nab, h7 = hip.sdf_nabla_fwd(surf, x, 3.0, precision=4)
    rgb = hip.radiance_fwd(rad, 1, x, view.to(DEV), nab, h7, precision=4).cpu().numpy()
    em.TERM = "fp16"
    try:
        parts = [em.emul_sdf_grad_bf16(surf_cpu, pts[i:i + 16].numpy(), 3.0) for i in range(0, 64, 16)]
    finally:
        em.TERM = "bf16"
    e_sdf, e_nab, e_h7 = (np.concatenate([p[k] for p in parts]) for k in range(3))
    # same data flow, but 11-bit activations make the arithmetic CHAOTIC at its own resolution: a 1e-7 difference of a pre-activation
    # (summation order; the GPU's fold of weight_norm) flips an activation to the neighbouring fp16 value (2.4e-4 relative) - most
    # points agree to fp32 noise, the rest to the arithmetic's resolution (measured: 22 % of the points, max 3.9e-4)
    np.testing.assert_array_equal(s_k2, sdf.cpu().numpy())                       # K2 and the reverse-mode kernel's forward sweep: same bits
    d_s = np.abs(sdf.cpu().numpy() - e_sdf)
    assert (d_s < 2e-5).mean() >= 0.6 and d_s.max() < 1.5e-3, (float((d_s < 2e-5).mean()), float(d_s.max()))
    np.testing.assert_allclose(nab.cpu().numpy(), e_nab, atol=5e-3, rtol=5e-3)
    np.testing.assert_allclose(h7.cpu().numpy(), e_h7, atol=2e-3)
    s_ref, n_ref, feat_ref = nets.surface_forward_with_nablas(sd, pts)
    d_bg = 3.0 - pts.norm(dim=-1)
    s_ref = torch.where(d_bg < s_ref, d_bg, s_ref)
    print(f"  fp16x2 vs oracle: sdf {np.abs(sdf.cpu().numpy() - s_ref.numpy()).max():.2e}, nabla {np.abs(nab.cpu().numpy() - n_ref.numpy()).max():.2e}")
    np.testing.assert_allclose(sdf.cpu().numpy(), s_ref.numpy(), atol=3e-3)
    np.testing.assert_allclose(nab.cpu().numpy(), n_ref.numpy(), atol=2e-2, rtol=2e-2)
    ref = nets.radiance_forward(sd, pts, view, n_ref, feat_ref, -1, -1).numpy()
    np.testing.assert_allclose(rgb, ref, atol=1e-2)


@pytest.mark.parametrize("fw", ["VolSDF", "NeuS"])
def test_fp16x2_renders_a_frame_close_to_fp32(fw):
    """Both renderers at precision 4 against the exact-fp32 HIP frame: a valid rendering of the same field (PSNR, finite, chunk
    invariant) - NOT the 1e-3 pixel contract, which this arithmetic is measured against in tools/parity_table.py."""
    from nerfart_amd import scene, rend_util
    H, W = 96, 54
    c2w, K = scene.camera(H, W)
    o, d, _ = rend_util.get_rays(c2w[None].to(DEV), K[None].to(DEV), H, W)
    out = {}
    for precision in ("fp32", "fp16x2"):
        model, rk, fn = scene.build_model(fw, seed=0, beta=0.01 if fw == "VolSDF" else None, device=DEV, precision=precision)
        kw = {k: v for k, v in rk.items() if k != "rayschunk"}
        extra = dict(require_nablas=True) if fw == "VolSDF" else {}
        rgb, depth, _ = fn(o, d, calc_normal=True, detailed_output=False, **extra, **kw)
        rgb2, _, _ = fn(o, d, calc_normal=True, detailed_output=False, rayschunk=1777, honor_rayschunk=True, **extra, **kw)
        assert torch.equal(rgb, rgb2) and torch.isfinite(rgb).all()
        out[precision] = rgb[0]
    err = (out["fp16x2"] - out["fp32"]).abs().max(dim=-1).values
    psnr = float(-10 * torch.log10(((out["fp16x2"] - out["fp32"]) ** 2).mean().clamp_min(1e-20)))
    print(f"  {fw} fp16x2 vs fp32, {H * W} rays: {int((err > 1e-3).sum())} past 1e-3, max {float(err.max()):.2e}, PSNR {psnr:.1f} dB")
    assert psnr > 55.0 and float(err.max()) < 0.1


def test_training_refuses_fp16x2():
    from nerfart_amd import scene
    from nerfart_amd.trainer import Trainer
    model, rk, fn = scene.build_model("VolSDF", seed=0, beta=0.01, device=DEV, precision="fp16x2")
    with pytest.raises(RuntimeError, match="bf16x3"):
        Trainer(model).native


def test_staged_renderer_equals_the_fused_one_and_the_mixed_sampler_mode_renders():
    """hip.volsdf_render_mixed (the per-stage entry points in the fused renderer's order) with the sampler on the SAME blob and
    precision reproduces nerfart_volsdf_render_fwd bit for bit, every output; with the sampler at precision 4
    (model.set_sampler_precision("fp16x2")) the final samples stay split-bf16: sdf / nabla / radiance at the chosen depths are
    the bf16x3 kernels' own values, and the frame is close to the fused bf16x3 frame."""
    from nerfart_amd import hip, scene, rend_util
    model, rk, render_fn = scene.build_model("VolSDF", seed=0, beta=0.01, device=DEV, precision="bf16x3")
    H, W = 48, 27
    c2w, K = scene.camera(H, W)
    o, d, _ = rend_util.get_rays(c2w[None].to(DEV), K[None].to(DEV), H, W)
    surf, rad = model.packed()
    alpha, beta = (float(t) for t in model.forward_ab())
    kw = dict(near=0.0, far=6.0, R_bg=3.0, alpha=alpha, beta=beta, max_upsample_steps=6, detailed=True, precision=1)
    fused = hip.volsdf_render(surf, rad, 1, o[0].contiguous(), d[0].contiguous(), **kw)
    staged = hip.volsdf_render_mixed(surf, rad, surf, 1, 1, o[0].contiguous(), d[0].contiguous(), **kw)
    assert set(fused) == set(staged)
    for k in fused:
        assert torch.equal(fused[k], staged[k]), k
    # ... and with the sampler on the fp16 blob at precision 4: the C entry point nerfart_volsdf_render_mixed_fwd against the same stages one by one
    model.set_sampler_precision("fp16x2")
    samp = model.packed_sampler()
    fused4 = hip.volsdf_render(surf, rad, 1, o[0].contiguous(), d[0].contiguous(), sampler=samp, **kw)
    staged4 = hip.volsdf_render_mixed(surf, rad, samp[0], samp[1], 1, o[0].contiguous(), d[0].contiguous(), **kw)
    for k in fused4:
        assert torch.equal(fused4[k], staged4[k]), k
    assert not torch.equal(fused4["d_vals"], fused["d_vals"])
    model.set_sampler_precision(None)
    rkk = {k: v for k, v in rk.items() if k != "rayschunk"}
    rgb_f, _, _ = render_fn(o, d, require_nablas=True, calc_normal=True, detailed_output=False, **rkk)
    model.set_sampler_precision("fp16x2")
    rgb_m, depth_m, ex = render_fn(o, d, require_nablas=True, calc_normal=True, detailed_output=True, **rkk)
    model.set_sampler_precision(None)
    # the final samples are evaluated by the split-bf16 kernels at the depths the cheaper sampler chose
    pts, _ = hip.ray_points(o[0].contiguous(), hip.normalize_dirs(d[0].contiguous()), ex["d_vals"][0].contiguous())
    sdf, nab, _ = hip.sdf_nabla_fwd(surf, pts, 3.0, precision=1)
    assert float((sdf.reshape(H * W, -1) - ex["implicit_surface"][0]).abs().max()) < 2e-5           # (precision 4 would be off by ~4e-4 here)
    assert float((nab.reshape(H * W, -1, 3) - ex["implicit_nablas"][0]).abs().max()) < 1e-4
    err = (rgb_m - rgb_f).abs().max(dim=-1).values
    print(f"  bf16x3 with the fp16x2 sampler vs fused bf16x3, {H * W} rays: {int((err > 1e-3).sum())} past 1e-3, max {float(err.max()):.2e}")
    assert float(err.max()) < 2e-2 and float((err > 1e-3).float().mean()) < 0.02


def test_finetune_step_with_the_fp16x2_sampler():
    """Training with model.set_sampler_precision("fp16x2"): pass 1's sampler (no gradient, volsdf.py:479) runs on the 2-MFMA kernels, the
    kept per-sample state and the whole pass 2 stay split-bf16; image and gradients stay close to the all-bf16x3 step (the samples move by
    the sampler's arithmetic only)."""
    from nerfart_amd import scene, rend_util
    from nerfart_amd.trainer import Trainer
    H, W = 24, 16
    c2w, K = scene.camera(H, W)
    o, d, _ = rend_util.get_rays(c2w[None].to(DEV), K[None].to(DEV), H, W)
    target = torch.rand(1, H * W, 3, generator=torch.Generator().manual_seed(1)).to(DEV)
    loss_fn = lambda pred, gt: ((pred - gt) ** 2).mean()
    res = {}
    for samp in (None, "fp16x2"):
        model, rk, render_fn = scene.build_model("VolSDF", seed=0, beta=0.01, device=DEV, precision="bf16x3")
        model.set_sampler_precision(samp)
        model.zero_grad()
        out = Trainer(model, pass2_rays=64).finetune_step(render_fn, o, d, target, H, loss_fn, **rk)
        res[samp] = (out["rgb"], {n: p.grad.clone() for n, p in model.named_parameters()})
    assert float((res[None][0] - res["fp16x2"][0]).abs().max()) < 2e-2
    for n, g in res[None][1].items():
        rel = float((res["fp16x2"][1][n] - g).norm() / (g.norm() + 1e-12))
        assert rel < 0.1, (n, rel)


def test_fp16x1_sampler_only_precision():
    """C-ABI precision 5 ("fp16x1", csrc/mlp_chain_f16x1.hip): K2 with ONE MFMA per product on the precision-4 blob - an opt-in arithmetic of Algorithm
    1's no-gradient SDF queries, measured (profiles/r09_guard_sweep_fp16x1_8views.json: - 27 % of K2, - 11 % of a frame at guard 0.05) and NOT shipped
    (one or two rays of 2,048 more than pure split-bf16 past 1e-3 on 2 of 8 views).  Held: close to the 2-MFMA kernel on the same blob (the
    dropped term is 2^-12 of a product), refused everywhere a value that reaches a pixel is computed, and a guarded frame with it is a valid
    rendering whose escalated rays are pure split-bf16's."""
    from nerfart_amd import hip, scene, rend_util
    model, rk, fn = scene.build_model("VolSDF", seed=0, beta=0.01, device=DEV, precision="bf16x3")
    g, v, b = model._surface_layers()
    blob4 = hip.pack_surface_blob(4, 6, g, v, b)
    model.set_sampler_precision("fp16x1", guard=0.05)                       # nearest-rounded one-term weights in the kernel's scaled softplus recursion
    blob5, p5 = model.packed_sampler()
    assert p5 == 5
    gen = torch.Generator().manual_seed(5)
    x = ((torch.rand(1 << 16, 3, generator=gen) * 2 - 1) * 1.5).to(DEV)
    s4 = hip.sdf_fwd(blob4, x, 3.0, precision=4)
    s5 = hip.sdf_fwd(blob5, x, 3.0, precision=5)
    with pytest.raises(hip.NerfartHipError, match="packed as"):
        hip.sdf_fwd(blob4, x[:256], 3.0, precision=5)                         # the 2-MFMA blob is NOT the 1-MFMA kernel's (its recursion is scaled): refused
    with pytest.raises(hip.NerfartHipError, match="packed as"):
        hip.sdf_fwd(blob5, x[:256], 3.0, precision=4)
    s1 = hip.sdf_fwd(model.packed()[0], x, 3.0, precision=1)
    e45, e51, e41 = (s5 - s4).abs(), (s5 - s1).abs(), (s4 - s1).abs()
    print(f"  fp16x1 vs fp16x2: max {float(e45.max()):.2e} mean {float(e45.mean()):.2e}; vs bf16x3: fp16x1 {float(e51.max()):.2e} / {float(e51.mean()):.2e}, fp16x2 {float(e41.max()):.2e} / {float(e41.mean()):.2e}")
    assert float(e45.max()) < 2e-3 and float(e45.mean()) < 2.5e-4 and float(e51.max()) < 2e-3 and float(e51.mean()) < 2e-4      # measured 9.6e-4 / 1.2e-4 (both sides carry their own 11-bit activation noise), 8.5e-4 / 1.0e-4
    with pytest.raises(hip.NerfartHipError, match="precision 5"):
        hip.sdf_nabla_fwd(blob5, x[:256], 3.0, precision=5)
    with pytest.raises(hip.NerfartHipError):
        hip.sdf_fwd(model.packed()[0], x[:256], 3.0, precision=5)           # a split-bf16 blob: the library's encoding check
    with pytest.raises(ValueError):
        model.set_precision("fp16x1")
    H, W = 96, 54
    c2w, K = scene.camera(H, W)
    o, d, _ = rend_util.get_rays(c2w[None].to(DEV), K[None].to(DEV), H, W)
    kw = dict(require_nablas=True, calc_normal=True, detailed_output=True, **rk)
    base, _, ex0 = fn(o, d, **kw)
    assert model.mode == "bf16x3+fp16x1 sampler (guard 0.05)"
    model.render_stats = {}
    rgb, _, ex = fn(o, d, **kw)
    err = (rgb - base).abs().max(dim=-1).values[0]
    same = (ex["iter_usage"] == ex0["iter_usage"])[0]
    print(f"  bf16x3 + fp16x1 sampler (guard 0.05) vs bf16x3, {H * W} rays: {int((err > 1e-3).sum())} past 1e-3, max {float(err.max()):.2e}, identical rounds "
          f"{float(same.float().mean()):.4f}, sampled twice {model.render_stats['escalated'] / model.render_stats['rays']:.3f}")
    assert torch.isfinite(rgb).all() and float(err.max()) < 2e-2 and int((err > 1e-3).sum()) <= H * W // 100 and float(same.float().mean()) > 0.97
    never = (ex["iter_usage"] < 0)[0]
    assert torch.equal(rgb[0][never], base[0][never]), "a never-converged ray is sampled again at the model's precision: pure split-bf16's pixel"
