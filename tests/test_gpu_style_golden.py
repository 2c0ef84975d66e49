"""The hand-written style-loss path on the MI355X (resampling gathers -> native CLIP ViT-B/32 encoder -> one-launch loss heads,
csrc/style_heads.hip + clip_vit.hip; the VGG16 perceptual term of csrc/vgg_conv.hip) against vectors the REFERENCE's own
criteria/*.py and Trainer.calc_style_loss produced on the CPU in fp32 (tests/golden/make_golden_style.py -> style_golden.npz).
Same seeded random-weight CLIP / VGG (fp16 on the GPU as `clip.load(..., device="cuda")` makes it); through the C ABI."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import style_inputs as si
from conftest import state_checksum

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "style_golden.npz")


@pytest.fixture(scope="module")
def sg():
    z = np.load(GOLDEN, allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def feats(sg):
    from nerfart_amd import criteria, clip_vit
    cpu = clip_vit.build_clip("cpu", seed=0)
    assert state_checksum(cpu.state_dict()) == str(sg["clip_state_sha256"])
    return criteria.ClipFeatures(model=clip_vit.build_clip(DEV, seed=0), device=DEV, synthetic=True, native=True)


def images(sg, name):
    H, W, _, _ = si.CASES[name]
    rgb, rgb_gt = si.image_pair(name)
    assert si.sha(rgb) == str(sg[name + "_rgb_sha256"]) and si.sha(rgb_gt) == str(sg[name + "_rgb_gt_sha256"])
    img = lambda t: t.reshape(1, H, W, 3).permute(0, 3, 1, 2).contiguous().to(DEV)
    return img(rgb), img(rgb_gt)


@pytest.mark.parametrize("name", list(si.CASES))
def test_resample_chain_matches_reference_preprocessing(sg, name):
    """nerfart_resample_fwd stages vs the reference's torchvision chains (clip_loss.py:166-168, contrastive_loss.py:98-101,
    patchnce_loss.py:98-117,211-215)."""
    from nerfart_amd import style_native as sn
    pred, _ = images(sg, name)
    H, W, target_hw, downscale = si.CASES[name]
    t = lambda k: torch.from_numpy(sg[name + "_" + k])
    st = lambda x: x[..., ::4, ::4].cpu()
    norm, norm_half = sn.normalize_affine(DEV, False), sn.normalize_affine(DEV, True)
    np.testing.assert_allclose(st(sn.resample(pred, (224, 224), mode="bicubic", affine=norm)).numpy(), t("pre_clip").numpy(), atol=2e-5)
    rh, rw = sn._short_side(H, W, 224)
    cc = [(int(round((rh - 224) / 2.0)), int(round((rw - 224) / 2.0)))]
    got = sn.resample(pred, (224, 224), resized_hw=(rh, rw), mode="bicubic", crops=cc, affine=norm_half)
    np.testing.assert_allclose(st(got).numpy(), t("pre_contrastive").numpy(), atol=2e-5)
    canvas = sn.resample(pred, target_hw, mode="bicubic", pad=(270, 270, 480, 480))
    np.testing.assert_allclose(st(canvas).numpy(), t("patchnce_canvas").numpy(), atol=2e-5)
    th = 224 if downscale == 1 else 112
    wins = [(int(r[0]), int(r[1]), th, th) for r in sg[name + "_draw_crops"][:2]]
    got = sn.resample(canvas, (224, 224), mode="bicubic", windows=wins, affine=norm_half)
    for n in range(2):
        np.testing.assert_allclose(st(got[n:n + 1]).numpy(), t(f"pre_patchnce_{n}").numpy(), atol=3e-5)


@pytest.mark.parametrize("name", list(si.CASES))
def test_native_style_loss_matches_reference_calc_style_loss(sg, feats, name):
    """Every CLIP term, the total and d total / d rgb of the reference's calc_style_loss (fp32 CPU) vs the kernels (fp16 encoder):
    loss <= 2e-3 relative, pixel gradient <= 1e-2 relative."""
    from nerfart_amd import criteria
    pred, gt = images(sg, name)
    H, W, target_hw, downscale = si.CASES[name]
    style = criteria.StyleLoss(feats, target_hw, src_text=si.SRC_TEXT, target_text=si.TARGET_TEXT, neg_texts=[str(t) for t in sg[name + "_neg_texts"]],
                               w_clip=si.WEIGHTS["w_clip"], w_contrastive=si.WEIGHTS["w_contrastive"], w_patchnce=si.WEIGHTS["w_patchnce"],
                               is_full_res=(downscale == 1), seed=si.DRAW_SEED, perceptual=None, w_perceptual=0.0)
    x = pred.clone().requires_grad_(True)
    total = style(x, gt)
    con_text, nce_texts, crops = style.last_draw
    assert con_text == str(sg[name + "_draw_contrastive_text"]) and nce_texts == [str(s) for s in sg[name + "_draw_patchnce_texts"]]
    assert [tuple(c) for c in crops] == [(int(r[0]), int(r[1])) for r in sg[name + "_draw_crops"]]
    parts = style.last_parts.cpu()
    ref_parts = [float(sg[name + "_loss_" + k]) for k in ("clip", "contrastive", "patchnce")]
    ref_total = sum(w * v for w, v in zip((si.WEIGHTS["w_clip"], si.WEIGHTS["w_contrastive"], si.WEIGHTS["w_patchnce"]), ref_parts))
    total.backward()
    g = x.grad.permute(0, 2, 3, 1).reshape(-1).cpu()
    idx = si.grad_sample_index(name)
    ref_g = sum(w * torch.from_numpy(sg[name + "_gradsample_" + k]) for w, k in
                zip((si.WEIGHTS["w_clip"], si.WEIGHTS["w_contrastive"], si.WEIGHTS["w_patchnce"]), ("clip", "contrastive", "patchnce")))
    rel = float((g[idx] - ref_g).norm() / ref_g.norm())
    print(f"  {name}: parts {[round(float(p), 6) for p in parts[1:]]} vs reference {[round(v, 6) for v in ref_parts]}; "
          f"total {float(total):.6f} vs {ref_total:.6f}; sampled pixel gradient rel diff {rel:.3e}")
    np.testing.assert_allclose(parts[1:].numpy(), ref_parts, rtol=2e-3)
    np.testing.assert_allclose(float(total), ref_total, rtol=2e-3)
    assert rel < 1e-2, rel


@pytest.mark.parametrize("name", ["cfg3", "square"])
def test_native_vgg_term_matches_reference_perp_loss(sg, name):
    """nerfart_vgg16_l1_fwd / _bwd vs criteria/perp_loss.py on the same weights."""
    from nerfart_amd import vgg
    pred, gt = images(sg, name)
    m = vgg.VGGPerceptualLoss(seed=0).to(DEV)
    x = pred.clone().requires_grad_(True)
    loss = m(x, gt)
    loss.backward()
    g = x.grad.permute(0, 2, 3, 1).reshape(-1).cpu()
    ref = torch.from_numpy(sg[name + "_gradsample_perceptual"])
    got = g[si.grad_sample_index(name)]
    rel = float((got - ref).norm() / ref.norm())
    cos = float(F.cosine_similarity(got, ref, dim=0))
    print(f"  {name}: perceptual {float(loss):.7f} vs reference {float(sg[name + '_loss_perceptual']):.7f}; sampled gradient rel {rel:.3e}, cosine {cos:.6f}")
    np.testing.assert_allclose(float(loss), float(sg[name + "_loss_perceptual"]), rtol=2e-3)
    assert rel < 2e-2 and cos > 0.9998, (rel, cos)
