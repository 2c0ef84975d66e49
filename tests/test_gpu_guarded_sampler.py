"""The GUARDED sampler of the shipped `mixed` mode (nerfart_volsdf_fine_sample_guarded / nerfart_volsdf_render_staged_fwd, round 6):
Algorithm 1 (volsdf.py:97-302) on the 2-MFMA kernels, with every ray whose outcome hangs on a marginal threshold decision - max B within guard * eps
of eps at a convergence check (:162-163, :240-242), or never converged (:294-300) - sampled again, from its first query, on the split-bf16 kernels.

Held here: the escalated rays' samples are BIT-IDENTICAL to a pure split-bf16 run; the others are bit-identical to the unguarded fp16x2 run or were
escalated; fused entry point == the stage entry points; perturb=True draws follow the rays through the compaction; edge cases (no rays, one ray,
everything escalated, nothing escalated); and the statistics the mode ships on, over 8 orbit views (test_gpu_configs.py holds the pixel budget)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(H=96, W=54, pose=3):
    from nerfart_amd import scene, rend_util, hip
    model, rk, render_fn = scene.build_model("VolSDF", seed=0, beta=0.01, device=DEV, precision="mixed")
    c2w, K = scene.camera(H, W, angle=scene.spiral(90)[pose])
    o, d, _ = rend_util.get_rays(c2w[None].to(DEV), K[None].to(DEV), H, W)
    o, d = o[0].contiguous(), d[0].contiguous()
    dn = hip.normalize_dirs(d)
    alpha, beta = (float(t.detach()) for t in model.forward_ab())
    return model, rk, render_fn, o, d, dn, alpha, beta


def _sample(model, o, dn, alpha, beta, blob, prec, escalate=None, guard=0.0, u=None, n_final=64, stats=None, max_iter=6, late_round=0):
    from nerfart_amd import hip
    return hip.volsdf_fine_sample(blob, o, dn, 0.0, 6.0, 3.0, alpha, beta, 0.1, 512, 512, n_final, max_iter, 10, precision=prec, u_final=u,
                                  escalate=escalate, guard=guard, stats=stats, late_round=late_round)


@pytest.mark.parametrize("perturb", [False, True])
def test_escalated_rays_are_the_split_bf16_run_bit_for_bit(perturb):
    model, rk, fn, o, d, dn, alpha, beta = _setup()
    surf, _ = model.packed()
    samp, sprec = model.packed_sampler()
    R = o.shape[0]
    u = torch.rand(R, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5)) if perturb else None
    pure = _sample(model, o, dn, alpha, beta, surf, 1, u=u)                                   # Algorithm 1 on the split-bf16 kernels
    cheap = _sample(model, o, dn, alpha, beta, samp, sprec, u=u)                              # ... on the 2-MFMA kernels, unguarded (round 5's mixed)
    for guard in (1e-6, 0.005, 0.05, 0.5):
        st = {}
        g = _sample(model, o, dn, alpha, beta, samp, sprec, escalate=(surf, 1), guard=guard, u=u, stats=st)
        like_pure = (g[0] == pure[0]).all(dim=1) & (g[1] == pure[1]) & (g[2] == pure[2])
        like_cheap = (g[0] == cheap[0]).all(dim=1) & (g[1] == cheap[1]) & (g[2] == cheap[2])
        assert bool((like_pure | like_cheap).all()), "a ray is either the cheap sampler's (every decision clear) or the split-bf16 run's, bit for bit"
        never = cheap[2] < 0
        assert bool(like_pure[never].all()), "every ray the cheap sampler could not converge is the split-bf16 run's"
        n_esc = int((~like_cheap).sum())
        assert n_esc <= st["escalated"] <= R and st["rays"] == R
        assert st["escalated"] >= int(never.sum())
        print(f"  guard {guard:g}, perturb {perturb}: {st['escalated']} of {R} rays sampled twice ({int(never.sum())} never converged on fp16x2); "
              f"rounds equal to the split-bf16 run's on {float((g[2] == pure[2]).float().mean()):.4f} (unguarded: {float((cheap[2] == pure[2]).float().mean()):.4f})")
    # a guard as wide as the decision itself sends (nearly) every undecided ray through the split-bf16 kernels
    st = {}
    g = _sample(model, o, dn, alpha, beta, samp, sprec, escalate=(surf, 1), guard=1e9, u=u, stats=st)
    assert st["escalated"] == R and all(torch.equal(a, b) for a, b in zip(g, pure)), "guard = inf: the whole batch is the split-bf16 run"
    # guard 0 / no escalation blob: the unguarded entry point
    g0 = _sample(model, o, dn, alpha, beta, samp, sprec, escalate=(surf, 1), guard=0.0, u=u)
    assert all(torch.equal(a, b) for a, b in zip(g0, cheap))


def test_guarded_sampler_edge_cases():
    model, rk, fn, o, d, dn, alpha, beta = _setup(H=24, W=16)
    surf, _ = model.packed()
    samp, sprec = model.packed_sampler()
    full = _sample(model, o, dn, alpha, beta, samp, sprec, escalate=(surf, 1), guard=0.005)
    # rays are independent: any sub-batch (one ray, a ragged tail, a permutation) gives the same rows
    for sel in (torch.tensor([7], device=DEV), torch.arange(0, 383, 3, device=DEV), torch.randperm(o.shape[0], device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))):
        part = _sample(model, o[sel].contiguous(), dn[sel].contiguous(), alpha, beta, samp, sprec, escalate=(surf, 1), guard=0.005)
        for a, b in zip(part, full):
            assert torch.equal(a, b[sel])
    empty = _sample(model, o[:0].contiguous(), dn[:0].contiguous(), alpha, beta, samp, sprec, escalate=(surf, 1), guard=0.005)
    assert empty[0].shape == (0, 64)
    # max_upsample_steps = 0: nothing can converge after the first check -> every undecided ray is "never converged" -> all of them escalate
    st = {}
    z = _sample(model, o, dn, alpha, beta, samp, sprec, escalate=(surf, 1), guard=0.005, stats=st, max_iter=0)
    zp = _sample(model, o, dn, alpha, beta, surf, 1, max_iter=0)
    assert torch.equal(z[2], zp[2]) and torch.equal(z[0][z[2] < 0], zp[0][zp[2] < 0])
    # 128 final samples per ray (the trainer's two draws from one run, Trainer.render_two_draws) go through the compaction too
    u = torch.rand(o.shape[0], 128, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
    two = _sample(model, o, dn, alpha, beta, samp, sprec, escalate=(surf, 1), guard=0.005, u=u, n_final=128)
    one = _sample(model, o, dn, alpha, beta, samp, sprec, escalate=(surf, 1), guard=0.005, u=u[:, :64].contiguous())
    assert torch.equal(two[0][:, :64], one[0]) and torch.equal(two[2], one[2])


@pytest.mark.parametrize("sampler", ["fp16x2", "fp16x1c"])
def test_late_round_rule_sends_the_branch_sensitive_rays_through_the_split_bf16_kernels(sampler):
    """nerfart_volsdf_fine_sample_guarded2's third rule (ABI 5; nets.DEFAULT_SAMPLER_LATE_ROUND = 3): a ray still active after round `late_round` is
    escalated there.  Every ray of the guarded run is then either the cheap sampler's own (it converged by round late_round with every decision
    clear) or the split-bf16 run's, bit for bit; every ray the split-bf16 run needs more than late_round rounds for - or never converges on - that
    the cheap run also carried past late_round is the split-bf16 run's; late_round = 0 is the two-rule sampler; the fused renderer passes it through."""
    from nerfart_amd import hip
    model, rk, fn, o, d, dn, alpha, beta = _setup()
    model.set_sampler_precision(sampler, guard=0.005, late_round=3)
    surf, rad = model.packed()
    samp, sprec = model.packed_sampler()
    assert sprec == (4 if sampler == "fp16x2" else 5)
    R = o.shape[0]
    pure = _sample(model, o, dn, alpha, beta, surf, 1)
    cheap = _sample(model, o, dn, alpha, beta, samp, sprec)
    two = _sample(model, o, dn, alpha, beta, samp, sprec, escalate=(surf, 1), guard=0.005)
    for late in (1, 3, 5):
        st = {}
        g = _sample(model, o, dn, alpha, beta, samp, sprec, escalate=(surf, 1), guard=0.005, late_round=late, stats=st)
        like_pure = (g[0] == pure[0]).all(dim=1) & (g[1] == pure[1]) & (g[2] == pure[2])
        like_cheap = (g[0] == cheap[0]).all(dim=1) & (g[1] == cheap[1]) & (g[2] == cheap[2])
        assert bool((like_pure | like_cheap).all())
        carried = (cheap[2] > late) | (cheap[2] < 0)                      # the cheap run did not decide these by round `late`
        assert bool(like_pure[carried].all()), "a ray the cheap sampler carries past late_round is the split-bf16 run's"
        assert bool((g[2][~like_pure] <= late).all()) and bool((g[2][~like_pure] >= 0).all()), "what stays the cheap sampler's converged by round late_round"
        assert st["escalated"] >= int(carried.sum())
        print(f"  {sampler}, late_round {late}: {st['escalated']} of {R} rays sampled twice ({int(carried.sum())} carried past round {late}); rounds equal to the "
              f"split-bf16 run's on {float((g[2] == pure[2]).float().mean()):.4f} (two rules: {float((two[2] == pure[2]).float().mean()):.4f})")
    g0 = _sample(model, o, dn, alpha, beta, samp, sprec, escalate=(surf, 1), guard=0.005, late_round=0)
    assert all(torch.equal(a, b) for a, b in zip(g0, two))
    kw = dict(near=0.0, far=6.0, R_bg=3.0, alpha=alpha, beta=beta, max_upsample_steps=6, detailed=True, precision=1)
    fused = hip.volsdf_render(surf, rad, 1, o, d, sampler=(samp, sprec), guard=0.005, late_round=3, **kw)
    staged = hip.volsdf_render_mixed(surf, rad, samp, sprec, 1, o, d, guard=0.005, late_round=3, **kw)
    for k in fused:
        assert torch.equal(fused[k], staged[k]), k
    rgb, _, ex = fn(o[None], d[None], require_nablas=True, calc_normal=True, detailed_output=True, **rk)      # the model's own call
    assert torch.equal(rgb[0], fused["rgb"]) and torch.equal(ex["iter_usage"][0], fused["iter_usage"])


def test_fused_staged_renderer_equals_the_stage_entry_points_with_the_guard_on():
    from nerfart_amd import hip
    model, rk, fn, o, d, dn, alpha, beta = _setup(H=48, W=27)
    surf, rad = model.packed()
    samp = model.packed_sampler()
    kw = dict(near=0.0, far=6.0, R_bg=3.0, alpha=alpha, beta=beta, max_upsample_steps=6, detailed=True, precision=1)
    st = {}
    fused = hip.volsdf_render(surf, rad, 1, o, d, sampler=samp, guard=0.02, stats=st, **kw)
    staged = hip.volsdf_render_mixed(surf, rad, samp[0], samp[1], 1, o, d, guard=0.02, **kw)
    assert st["escalated"] > 0
    for k in fused:
        assert torch.equal(fused[k], staged[k]), k
    # the model's own call: volume_render passes model.sampler_guard; the stats hook counts
    model.render_stats = {}
    rgb, _, ex = fn(o[None], d[None], require_nablas=True, calc_normal=True, detailed_output=True, **rk)
    assert model.render_stats["rays"] == o.shape[0] and model.render_stats["escalated"] > 0
    ref = hip.volsdf_render(surf, rad, 1, o, d, sampler=samp, guard=model.sampler_guard, late_round=model.sampler_late_round, **kw)
    assert torch.equal(rgb[0], ref["rgb"]) and torch.equal(ex["iter_usage"][0], ref["iter_usage"])
    # rays that never converge render exactly as in the pure split-bf16 mode
    model.set_precision("bf16x3")
    rgb_p, _, ex_p = fn(o[None], d[None], require_nablas=True, calc_normal=True, detailed_output=True, **rk)
    never = ex_p["iter_usage"][0] < 0
    model.set_precision("mixed")
    never_m = ex["iter_usage"][0] < 0
    both = never & never_m
    assert int(both.sum()) > 0 and torch.equal(rgb[0][both], rgb_p[0][both])


def test_radiance_net_at_its_own_precision_changes_nothing_but_the_radiance():
    """nerfart_volsdf_render_staged_fwd's rad_precision (VERDICT r05 next 3 i): same samples, same sdf / nabla, the radiance MLP on the 2-MFMA kernels;
    measured 1.0e-4 max on a pixel (profiles/r08_radiance_precision.json) - an opt-in (model.set_radiance_precision), not the shipped arithmetic."""
    model, rk, fn, o, d, dn, alpha, beta = _setup(H=48, W=27)
    a_rgb, _, a = fn(o[None], d[None], require_nablas=True, calc_normal=True, detailed_output=True, **rk)
    model.set_radiance_precision("fp16x2")
    b_rgb, _, b = fn(o[None], d[None], require_nablas=True, calc_normal=True, detailed_output=True, **rk)
    model.set_radiance_precision(None)
    for k in ("d_vals", "implicit_surface", "implicit_nablas", "iter_usage", "beta_map"):
        assert torch.equal(a[k], b[k]), k
    assert not torch.equal(a["radiance"], b["radiance"])
    e = (a_rgb - b_rgb).abs().max()
    print(f"  radiance net fp16x2 vs split-bf16, same samples: max pixel difference {float(e):.2e}")
    assert float(e) < 3e-4
    c_rgb, _, _ = fn(o[None], d[None], require_nablas=True, calc_normal=True, detailed_output=False, **rk)
    assert torch.equal(c_rgb, a_rgb), "set_radiance_precision(None) restores the model's arithmetic"


def test_abi3_mixed_entry_point_is_the_staged_one_with_the_guard_off():
    """nerfart_volsdf_render_mixed_fwd (ABI 3) keeps its signature and meaning: = nerfart_volsdf_render_staged_fwd(rad_precision = precision, guard = 0)."""
    from nerfart_amd import hip
    model, rk, fn, o, d, dn, alpha, beta = _setup(H=24, W=16)
    surf, rad = model.packed()
    samp, sprec = model.packed_sampler()
    R, ns, ni = o.shape[0], 128, 64
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=DEV)
    rgb, depth, acc, usage = f(R, 3), f(R), f(R), f(R)
    nb = hip.lib.nerfart_volsdf_render_workspace_bytes(R, ns, ni, 6, 8192)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    lt = lambda n: hip.lin_table(n, o.device).data_ptr()
    rc = hip.lib.nerfart_volsdf_render_mixed_fwd(surf.data_ptr(), rad.data_ptr(), 1, samp.data_ptr(), sprec, 1, o.data_ptr(), d.data_ptr(), R, 0.0, 6.0, 3.0, alpha, beta, 0.1,
                                                 ns, ni, 6, 10, 0, 8192, lt(ns), lt(4 * ns), lt(4 * ns + 2), lt(ni), 0, rgb.data_ptr(), depth.data_ptr(), acc.data_ptr(),
                                                 None, None, None, None, None, None, None, None, None, usage.data_ptr(), ws.data_ptr(), ws.numel(),
                                                 torch.cuda.current_stream().cuda_stream)
    assert rc == 0, hip.lib.nerfart_last_error()
    ref = hip.volsdf_render(surf, rad, 1, o, d, near=0.0, far=6.0, R_bg=3.0, alpha=alpha, beta=beta, max_upsample_steps=6, detailed=True, precision=1,
                            sampler=(samp, sprec), guard=0.0)
    assert torch.equal(rgb, ref["rgb"]) and torch.equal(depth, ref["depth_volume"]) and torch.equal(usage, ref["iter_usage"])


_SCAN_CHILD = r'''
import sys, time, torch
sys.path.insert(0, %r)
from nerfart_amd import scene, rend_util, hip
model, rk, fn = scene.build_model("VolSDF", seed=0, beta=0.01, device="cuda", precision="mixed")
H, W = 480, 270
out = {}
for pose in (5, 40):
    c2w, K = scene.camera(H, W, angle=scene.spiral(90)[pose])
    o, d, _ = rend_util.get_rays(c2w[None].cuda(), K[None].cuda(), H, W)
    dn = hip.normalize_dirs(d[0].contiguous())
    alpha, beta = (float(t.detach()) for t in model.forward_ab())
    sa = model.sampler_args()
    run = lambda: hip.volsdf_fine_sample(sa["blob"], o[0].contiguous(), dn, 0.0, 6.0, 3.0, alpha, beta, 0.1, 512, 512, 64, 6, 10, precision=sa["precision"],
                                         escalate=sa["escalate"], guard=sa["guard"], late_round=sa["late_round"])
    run(); torch.cuda.synchronize(); t0 = time.perf_counter()
    r = run(); torch.cuda.synchronize()
    out[f"ms_{pose}"] = (time.perf_counter() - t0) * 1e3
    out[f"pose_{pose}"] = tuple(t.cpu() for t in r)
# the up-sampling stage on its own (it writes every interval's bound: the w_out / clamp path of the scan), 2,048 rays at three sample counts
g = torch.Generator().manual_seed(3)
for n in (512, 1024, 1536, 2048):
    dA = (torch.rand(2048, n, generator=g) * 6).sort(-1)[0].cuda()
    sA = (torch.randn(2048, n, generator=g) * 0.3).cuda()
    act = torch.arange(2048, dtype=torch.int32).cuda()
    bp = (torch.rand(2048, generator=g) * 0.05 + 0.005).cuda()
    d_new = torch.empty(2048, 512, device="cuda")
    hip._check(hip.lib.nerfart_volsdf_upsample(2048, n, n, 512, dA.data_ptr(), sA.data_ptr(), act.data_ptr(), bp.data_ptr(), hip.lin_table(514, dA.device).data_ptr(), 1,
                                               d_new.data_ptr(), torch.cuda.current_stream().cuda_stream), "upsample")
    out[f"up_{n}"] = d_new.cpu()
torch.save(out, sys.argv[1])
'''


def test_cached_scan_equals_generic_scan(tmp_path):
    """csrc/ray_common.h (round 6): the register-cached error-bound scan (segments of <= 24 intervals per lane: the first check and rounds 1 - 2) against the
    generic two-pass form (the test-only variant library nerfart_amd.build builds with -DNERFART_SCAN_GENERIC): the guarded sampler's outputs for two
    whole 480 x 270 frames and the up-sampling stage at 512 .. 2,048 samples per ray, BIT for bit."""
    import os
    import subprocess
    import sys
    from conftest import REPO
    from nerfart_amd import build
    vlib = build.variant_lib("scan_generic")
    assert os.path.exists(vlib), "python -m nerfart_amd.build builds the test-only variant library"
    res = {}
    for name, lib in (("cached", None), ("generic", vlib)):
        env = dict(os.environ)
        if lib:
            env["NERFART_HIP_LIB"] = lib
        f = str(tmp_path / f"{name}.pt")
        r = subprocess.run([sys.executable, "-c", _SCAN_CHILD % REPO, f], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[name] = torch.load(f)
    for k in res["cached"]:
        if k.startswith("ms_"):
            continue
        a, b = res["cached"][k], res["generic"][k]
        if isinstance(a, tuple):
            assert all(torch.equal(x, y) for x, y in zip(a, b)), k
        else:
            assert torch.equal(a, b), k
    print("  guarded sampler of a 480x270 frame, ms: " + ", ".join(f"{k[3:]}: cached {res['cached'][k]:.1f} / generic {res['generic'][k]:.1f}" for k in res["cached"] if k.startswith("ms_")))
