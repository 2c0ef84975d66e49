"""The image side of the style losses on the hand-written kernels (rows a20-a22; csrc/style_heads.hip) against the oracle: the
torch restatement of the reference's torchvision preprocessing chains and loss heads (nerfart_amd.criteria, pinned on the CPU in
tests/test_clip.py), evaluated in fp32 on the CPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _cmp(name, got, ref, atol):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    err = (got - ref).abs().max().item()
    print(f"  [{name}] max abs err {err:.3e} (ref max {ref.abs().max().item():.3e})")
    assert got.shape == ref.shape and err <= atol, (name, err)


def test_resample_stages_match_torch_chains():
    from nerfart_amd import criteria, style_native as sn
    g = torch.Generator().manual_seed(0)
    for (H, W) in ((480, 270), (120, 68), (64, 96)):
        x = torch.rand(1, 3, H, W, generator=g)
        cot = torch.randn(1, 3, 224, 224, generator=g)
        xd = x.to(DEV)
        # CLIPLoss.preprocess
        a = x.clone().requires_grad_(True)
        ref = criteria._normalize(criteria.resize(a, (224, 224), "bicubic"))
        (ref * cot).sum().backward()
        b = xd.clone().requires_grad_(True)
        got = sn.resample(b, (224, 224), mode="bicubic", affine=sn.normalize_affine(DEV, False))
        (got * cot.to(DEV)).sum().backward()
        _cmp(f"{H}x{W} resize 224 bicubic + normalize", got, ref, 2e-5)
        _cmp(f"{H}x{W}   ... gradient", b.grad, a.grad, 2e-5 * max(1.0, a.grad.abs().max().item()))
        # ContrastiveLoss.preprocess
        a = x.clone().requires_grad_(True)
        ref = criteria._normalize(criteria.center_crop(criteria.resize((a + 1.0) / 2.0, 224, "bicubic"), 224))
        (ref * cot).sum().backward()
        rh, rw = sn._short_side(H, W, 224)
        cc = [(int(round((rh - 224) / 2.0)), int(round((rw - 224) / 2.0)))]
        b = xd.clone().requires_grad_(True)
        got = sn.resample(b, (224, 224), resized_hw=(rh, rw), mode="bicubic", crops=cc, affine=sn.normalize_affine(DEV, True))
        (got * cot.to(DEV)).sum().backward()
        _cmp(f"{H}x{W} (x+1)/2, resize short side, centre crop, normalize", got, ref, 2e-5)
        _cmp(f"{H}x{W}   ... gradient", b.grad, a.grad, 2e-5 * max(1.0, a.grad.abs().max().item()))
    # PatchNCE chain at the benchmark size: pad, resize to H x W, 112^2 crops up-sampled x2, (x+1)/2 + bilinear identity + normalize
    H, W = 480, 270
    x = torch.rand(1, 3, H, W, generator=g)
    crops = [(100, 0), (150, 158), (268, 77), (123, 100)]
    cot = torch.randn(len(crops), 3, 224, 224, generator=g)
    a = x.clone().requires_grad_(True)
    full = criteria.resize(F.pad(a, (270, 270, 480, 480)), (H, W), "bicubic")
    ref = torch.cat([criteria._normalize(criteria.resize((F.interpolate(full[..., i:i + 112, j:j + 112], size=(224, 224), mode="bicubic",
                                                                          align_corners=False) + 1.0) / 2.0, (224, 224), "bilinear")) for (i, j) in crops])
    (ref * cot).sum().backward()
    b = x.to(DEV).requires_grad_(True)
    fd = sn.resample(b, (H, W), mode="bicubic", pad=(270, 270, 480, 480))
    _cmp("pad + resize to H x W", fd, full, 2e-5)
    got = sn.resample(fd, (224, 224), mode="bicubic", windows=[(i, j, 112, 112) for (i, j) in crops], affine=sn.normalize_affine(DEV, True))
    (got * cot.to(DEV)).sum().backward()
    _cmp("crops x2 + normalize", got, ref, 3e-5)
    _cmp("  ... gradient through both stages", b.grad, a.grad, 3e-5 * max(1.0, a.grad.abs().max().item()))
    # bilinear mode with a real scale change
    a = torch.rand(2, 3, 37, 53, generator=g)
    ref = F.interpolate(a, size=(224, 224), mode="bilinear", align_corners=False)
    _cmp("bilinear 37x53 -> 224", sn.resample(a.to(DEV), (224, 224), mode="bilinear"), ref, 2e-6)
    ref = F.interpolate(a, size=(20, 31), mode="bilinear", align_corners=False)
    _cmp("bilinear 37x53 -> 20x31", sn.resample(a.to(DEV), (20, 31), mode="bilinear"), ref, 2e-6)


def test_heads_value_and_feature_gradient_match_autograd():
    from nerfart_amd import criteria, style_native as sn
    g = torch.Generator().manual_seed(1)
    P, S, T = 12, 8, 79
    feats = torch.randn(4 + P, 512, generator=g) * 3.0
    unit = lambda t: t / t.norm(dim=-1, keepdim=True)
    t_tgt, t_con, t_neg = unit(torch.randn(T, 512, generator=g)), unit(torch.randn(T, 512, generator=g)), unit(torch.randn(S, T, 512, generator=g))
    tdir = unit(torch.randn(1, 512, generator=g))
    w = (1.0, 0.2, 0.1)
    f = feats.clone().requires_grad_(True)
    fh = f / f.norm(dim=-1, keepdim=True)
    edit = fh[0:1] - fh[1:2].detach()
    edit = edit / edit.norm(dim=-1, keepdim=True)
    l_dir = (1.0 - F.cosine_similarity(edit, tdir)).mean()
    tg, sr = fh[2:3], fh[3:4].detach()
    l_con = torch.mean(F.pairwise_distance(tg, t_tgt, keepdim=True) ** 2 + torch.clamp(2.0 - F.pairwise_distance(tg, t_con, keepdim=True), min=0.0) ** 2
                       + torch.clamp(2.0 - F.pairwise_distance(tg, sr, keepdim=True), min=0.0) ** 2)
    fp = fh[4:, None, :]
    pos = torch.exp(F.cosine_similarity(fp, t_tgt[None], dim=-1) / 0.07)
    neg = sum(torch.exp(F.cosine_similarity(fp, t_neg[s][None], dim=-1) / 0.07) for s in range(S))
    l_nce = (-torch.log(pos / (pos + neg))).mean(dim=1).sum()
    total = w[0] * l_dir + w[1] * l_con + w[2] * l_nce
    total.backward()
    fd = feats.to(DEV).requires_grad_(True)
    tot, parts = sn._Heads.apply(fd, P, tdir.to(DEV).contiguous(), t_tgt.to(DEV).contiguous(), t_con.to(DEV).contiguous(), t_neg.to(DEV).contiguous(),
                                 w, 2.0, 0.07)
    tot.backward()
    print(f"  heads: total {float(tot):.6f} vs {float(total):.6f}; parts {parts.cpu().tolist()} vs {[float(total), float(l_dir), float(l_con), float(l_nce)]}")
    np.testing.assert_allclose(parts.cpu().numpy(), [float(total), float(l_dir), float(l_con), float(l_nce)], rtol=2e-5, atol=2e-6)
    ref_g = f.grad
    ref_g = torch.where(torch.tensor([i in (1, 3) for i in range(4 + P)])[:, None], torch.zeros_like(ref_g), ref_g)
    _cmp("d total / d feats", fd.grad, ref_g, 2e-5 * ref_g.abs().max().item() + 1e-8)


def test_style_loss_native_path_matches_torch_path():
    """criteria.StyleLoss end to end: kernels (resample -> native encoder -> heads) vs the torch modules on the same GPU model."""
    from nerfart_amd import criteria, clip_vit
    g = torch.Generator().manual_seed(4)
    H, W = 480, 270
    gt, pred = torch.rand(1, 3, H, W, generator=g).to(DEV), torch.rand(1, 3, H, W, generator=g).to(DEV)
    out = {}
    for native in (True, False):
        feats = criteria.ClipFeatures(model=clip_vit.build_clip(DEV, seed=0), device=DEV, synthetic=True, native=native,
                                      templates=["a photo of a {}.", "a sketch of a {}.", "art of the {}.", "a {} in a video game."])
        style = criteria.StyleLoss(feats, (H, W), neg_texts=[f"negative prompt {i}" for i in range(11)], seed=5)
        p = pred.clone().requires_grad_(True)
        v = style(p, gt)
        v.backward()
        out[native] = (float(v), p.grad.detach().float().cpu())
    rel = ((out[True][1] - out[False][1]).norm() / out[False][1].norm()).item()
    cos = F.cosine_similarity(out[True][1].flatten(), out[False][1].flatten(), dim=0).item()
    print(f"  StyleLoss native {out[True][0]:.5f} vs torch-fp16 path {out[False][0]:.5f}; pixel gradient rel diff {rel:.3e}, cosine {cos:.5f}")
    assert abs(out[True][0] - out[False][0]) <= 2e-2 * max(1.0, abs(out[False][0]))
    assert cos > 0.99
