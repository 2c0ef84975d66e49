"""csrc/pass2_operands.hip against the torch formulation it replaced (what autograd does between the network calls of the
reference's pass 2, models/frameworks/volsdf.py:759-770): sample points, the cotangents the second-order sweep starts from
(sphere clamp of volsdf.py:97-100, eikonal gradient of volsdf.py:764-768 with the per-patch mean) and the narrow bf16 operands
of the weight-gradient reductions (encoding of models/base.py:46-64 and its tangent, radiance inputs, sigmoid delta, [sbar; 1])."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0) if torch.cuda.is_available() else None


def _hip():
    from nerfart_amd import hip
    return hip


def _embed(x, multires):
    """models/base.py:46-64"""
    if multires < 0:
        return x
    out = [x]
    for k in range(multires):
        out += [torch.sin(x * 2.0 ** k), torch.cos(x * 2.0 ** k)]
    return torch.cat(out, dim=-1)


def _embed_tangent(x, d, multires):
    if multires < 0:
        return d
    out = [d]
    for k in range(multires):
        f = 2.0 ** k
        out += [torch.cos(x * f) * (d * f), -torch.sin(x * f) * (d * f)]
    return torch.cat(out, dim=-1)


def _check_operand(out, want, M, split):
    """out [rows, 64] bf16 against fp32 columns `want` [M, c]: hi parts = bf16(want) up to 1 bf16 ulp on rare entries (the kernel's
    sinf / cosf against torch's: a last-place difference next to a rounding boundary), hi + lo = want to 2^-16; everything else 0."""
    c = want.shape[1]
    o = out.float()
    hi = o[:M, :c]
    ref_hi = want.to(torch.bfloat16).float()
    exact = (hi == ref_hi).float().mean().item()
    assert exact > 0.999, exact
    ulp = torch.maximum(ref_hi.abs(), torch.full_like(ref_hi, 1e-30)) * 2.0 ** -7
    assert bool(((hi - ref_hi).abs() <= ulp).all())
    if split:
        assert c <= 32
        lo = o[:M, 32:32 + c]
        err = (hi + lo - want).abs()
        assert bool((err <= want.abs() * 2.0 ** -15 + 1e-7).all()), err.max().item()
        assert float(o[:M, c:32].abs().sum()) == 0.0 and float(o[:M, 32 + c:].abs().sum()) == 0.0
    else:
        assert float(o[:M, c:].abs().sum()) == 0.0
    assert float(o[M:].abs().sum()) == 0.0


def test_ray_points_are_torchs_points_bit_for_bit():
    hip = _hip()
    g = torch.Generator().manual_seed(0)
    R, P = 777, 192
    o = (torch.rand(R, 3, generator=g) * 2 - 1).to(DEV)
    dn = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(DEV)
    d = torch.sort(torch.rand(R, P, generator=g) * 6, dim=-1)[0].to(DEV)
    pts, view = hip.ray_points(o, dn, d)
    want = (o[:, None, :] + dn[:, None, :] * d[:, :, None]).reshape(-1, 3)
    assert torch.equal(pts, want)
    assert torch.equal(view, dn[:, None, :].expand(R, P, 3).reshape(-1, 3))
    pts2, none = hip.ray_points(o, dn, d, want_view=False)
    assert none is None and torch.equal(pts2, want)


@pytest.mark.parametrize("R,group,w", [(4800, 1200, 0.1), (2500, 1200, 0.1), (300, None, 0.1), (300, 1200, 0.0)])
def test_cotangents_match_the_torch_formulation(R, group, w):
    hip = _hip()
    from nerfart_amd import autodiff
    g = torch.Generator().manual_seed(1)
    P, Rbg = 64, 3.0
    M = R * P
    pts = (torch.rand(M, 3, generator=g) * 4.4 - 2.2).to(DEV)
    # a third of the points sit exactly on the clamp's other branch (sdf = R - |x|), the rest inside the net's
    net = (torch.rand(M, generator=g) * 0.5 - 0.25).to(DEV)
    sphere = Rbg - pts.norm(dim=-1)
    sdf = torch.minimum(net, sphere)
    g_sdf = torch.randn(M, generator=g).to(DEV)
    nab = (torch.randn(M, 3, generator=g) * 0.7).to(DEV)
    g_n = torch.randn(M, 3, generator=g).to(DEV) * 1e-3
    extra = torch.randn(M, 3, generator=g).to(DEV) * 1e-3
    sbar, nbar, eik_ray = hip.volsdf_pass2_cotangents(pts, sdf, g_sdf, nab, g_n, R, P, Rbg, w, group, g_n_extra=extra)
    clamped = sdf >= (Rbg - pts.norm(dim=-1)) - 1e-6
    assert 0.05 < clamped.float().mean().item() < 0.95
    assert torch.equal(sbar, torch.where(clamped, torch.zeros_like(sdf), g_sdf))
    want_n = g_n + extra
    want_e = torch.zeros((), device=DEV)
    if w:
        want_e, g_eik = autodiff._eikonal_terms(nab, R, P, w, group)
        want_n = g_n + g_eik + extra
    assert float((nbar - want_n).abs().max()) <= 1e-6 * float(want_n.abs().max()) + 1e-9
    assert abs(float(eik_ray.sum()) - float(want_e)) <= 2e-6 * abs(float(want_e)) + 1e-12


@pytest.mark.parametrize("multires,M", [(6, 1000), (4, 130), (-1, 64), (0, 77), (10, 129)])
def test_embed_pair_operand(multires, M):
    hip = _hip()
    g = torch.Generator().manual_seed(2)
    Mp = (M + 63) // 64 * 64
    pts = (torch.rand(M, 3, generator=g) * 4 - 2).to(DEV)
    d = (torch.randn(M, 3, generator=g) * 1e-2).to(DEV)
    out = hip.wgrad_operand_embed_pair(pts, d, Mp, multires)
    assert out.shape == (2 * Mp, 64) and out.dtype == torch.bfloat16
    c = 3 if multires < 0 else 3 + 6 * multires
    _check_operand(out[:Mp], _embed(pts, multires), M, c <= 32)
    _check_operand(out[Mp:], _embed_tangent(pts, d, multires), M, c <= 32)


@pytest.mark.parametrize("mx,mv,M", [(-1, -1, 500), (-1, 4, 129), (2, 1, 64)])
def test_radiance_input_operand(mx, mv, M):
    hip = _hip()
    g = torch.Generator().manual_seed(3)
    Mp = (M + 127) // 128 * 128
    x = (torch.rand(M, 3, generator=g) * 4 - 2).to(DEV)
    v = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=-1).to(DEV)
    n = torch.randn(M, 3, generator=g).to(DEV)
    out = hip.wgrad_operand_inputs(x, mx, v, mv, n, Mp)
    want = torch.cat([_embed(x, mx), _embed(v, mv), n], dim=-1)
    _check_operand(out, want, M, want.shape[1] <= 32)


def test_rgb_delta_and_sbar_operands():
    hip = _hip()
    g = torch.Generator().manual_seed(4)
    M, Mp = 1000, 1024
    rgb = torch.rand(M, 3, generator=g).to(DEV)
    g_rgb = torch.randn(M, 3, generator=g).to(DEV)
    out, b4, d4 = hip.wgrad_operand_rgb_delta(rgb, g_rgb, Mp, want_d4=True)
    want = g_rgb * rgb * (1.0 - rgb)
    assert float((d4 - want).abs().max()) <= 1e-7
    assert float((b4 - want.double().sum(0).float()).abs().max()) <= 1e-5 * float(want.abs().sum(0).max())
    assert hip.wgrad_operand_rgb_delta(rgb, g_rgb, Mp)[2] is None
    _check_operand(out, d4, M, True)
    sbar = torch.randn(M, generator=g).to(DEV)
    s = hip.wgrad_operand_sbar_ones(sbar, Mp)
    assert s.shape == (2 * Mp, 64)
    _check_operand(s[:Mp], sbar[:, None], M, True)
    ones = s[Mp:].float()
    assert bool((ones[:, 0] == 1.0).all()) and float(ones[:, 1:].abs().sum()) == 0.0


def test_empty_inputs():
    hip = _hip()
    z3 = torch.zeros(0, 3, device=DEV)
    pts, view = hip.ray_points(z3, z3, torch.zeros(0, 192, device=DEV))
    assert pts.shape == (0, 3) and view.shape == (0, 3)
    out = hip.wgrad_operand_embed_pair(z3, z3, 64, 6)                 # no points: the padded operand is all zero
    assert out.shape == (128, 64) and float(out.float().abs().sum()) == 0.0
    s = hip.wgrad_operand_sbar_ones(torch.zeros(0, device=DEV), 64).float()
    assert float(s[:64].abs().sum()) == 0.0 and bool((s[64:, 0] == 1.0).all())
    _, b4, _ = hip.wgrad_operand_rgb_delta(z3, z3, 128)
    assert b4.tolist() == [0.0, 0.0, 0.0]
    sbar, nbar, eik = hip.volsdf_pass2_cotangents(z3, torch.zeros(0, device=DEV), torch.zeros(0, device=DEV), z3, z3, 0, 192, 3.0, 0.1, 1200)
    assert sbar.numel() == 0 and nbar.shape == (0, 3) and eik.numel() == 0


def test_operand_argument_checks():
    hip = _hip()
    x = torch.zeros(10, 3, device=DEV)
    with pytest.raises(hip.NerfartHipError):
        hip.wgrad_operand_embed_pair(x, x, 8, 6)                      # rows_pad < M
    with pytest.raises(hip.NerfartHipError):
        hip.wgrad_operand_inputs(x, 10, x, 4, x, 64)                  # 63 + 27 + 3 columns do not fit 64
    with pytest.raises(hip.NerfartHipError):
        hip.wgrad_operand_embed_pair(x.cpu(), x, 64, 6)
