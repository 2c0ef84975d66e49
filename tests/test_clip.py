"""Rows a20-a23 (CLIP ViT-B/32 + loss heads), CPU.  PARITY UNPINNED against the OpenAI weights (no checkpoint, no
network): the architecture is pinned against `transformers.CLIPModel` (random weights, default config = ViT-B/32)
through the OpenAI -> HF parameter-name map of SURVEY.md 8c; the loss heads against closed-form restatements."""
import os

import numpy as np
import pytest
import torch

transformers = pytest.importorskip("transformers")


def _hf_and_mine():
    from nerfart_amd import clip_vit
    torch.manual_seed(0)
    hf = transformers.CLIPModel(transformers.CLIPConfig()).eval()
    mine = clip_vit.build_clip("cpu", seed=1)
    sd = {}
    h = hf.state_dict()
    sd["visual.conv1.weight"] = h["vision_model.embeddings.patch_embedding.weight"]
    sd["visual.class_embedding"] = h["vision_model.embeddings.class_embedding"]
    sd["visual.positional_embedding"] = h["vision_model.embeddings.position_embedding.weight"]
    for a, b in (("visual.ln_pre", "vision_model.pre_layrnorm"), ("visual.ln_post", "vision_model.post_layernorm"),
                 ("ln_final", "text_model.final_layer_norm")):
        sd[a + ".weight"], sd[a + ".bias"] = h[b + ".weight"], h[b + ".bias"]
    sd["visual.proj"] = h["visual_projection.weight"].T
    sd["text_projection"] = h["text_projection.weight"].T
    sd["token_embedding.weight"] = h["text_model.embeddings.token_embedding.weight"]
    sd["positional_embedding"] = h["text_model.embeddings.position_embedding.weight"]
    sd["logit_scale"] = h["logit_scale"]
    for pre, hpre in (("visual.transformer", "vision_model"), ("transformer", "text_model")):
        for i in range(12):
            o, m = f"{pre}.resblocks.{i}", f"{hpre}.encoder.layers.{i}"
            sd[o + ".attn.in_proj_weight"] = torch.cat([h[f"{m}.self_attn.{x}_proj.weight"] for x in "qkv"])
            sd[o + ".attn.in_proj_bias"] = torch.cat([h[f"{m}.self_attn.{x}_proj.bias"] for x in "qkv"])
            for a, b in ((".attn.out_proj", ".self_attn.out_proj"), (".ln_1", ".layer_norm1"), (".ln_2", ".layer_norm2"),
                         (".mlp.c_fc", ".mlp.fc1"), (".mlp.c_proj", ".mlp.fc2")):
                sd[o + a + ".weight"], sd[o + a + ".bias"] = h[m + b + ".weight"], h[m + b + ".bias"]
    mine.load_state_dict(sd)
    return hf, mine


def test_vitb32_image_and_text_encoders_match_transformers():
    from nerfart_amd import clip_vit
    hf, mine = _hf_and_mine()
    n_params = sum(p.numel() for p in mine.parameters())
    assert n_params == 151277313, n_params                                  # ViT-B/32: 87.85 M vision + 63.43 M text + proj + scale
    g = torch.Generator().manual_seed(3)
    img = torch.randn(2, 3, 224, 224, generator=g)
    with torch.no_grad():
        a = mine.encode_image(img)
        b = hf.visual_projection(hf.vision_model(pixel_values=img).pooler_output)
    np.testing.assert_allclose(a.numpy(), b.numpy(), atol=2e-4, rtol=1e-3)
    tok = torch.stack([clip_vit.synthetic_tokens("a photo of a face"), clip_vit.synthetic_tokens("painting, oil on canvas")])
    assert tok.shape == (2, 77) and (tok.argmax(-1) == torch.tensor([18, 24])).all()
    with torch.no_grad():
        a = mine.encode_text(tok)
        hid = hf.text_model(input_ids=tok).last_hidden_state                 # after final_layer_norm
        b = hf.text_projection(hid[torch.arange(2), tok.argmax(-1)])
    np.testing.assert_allclose(a.numpy(), b.numpy(), atol=2e-4, rtol=1e-3)


def test_image_encoder_backward_reaches_pixels():
    _, mine = _hf_and_mine()
    img = torch.rand(1, 3, 224, 224, requires_grad=True)
    mine.encode_image(img).square().sum().backward()
    assert img.grad is not None and torch.isfinite(img.grad).all() and float(img.grad.abs().max()) > 0


@pytest.fixture(scope="module")
def feats():
    from nerfart_amd import clip_vit, criteria
    return criteria.ClipFeatures(model=clip_vit.build_clip("cpu", seed=0), device="cpu", synthetic=True)


def test_preprocess_shapes_and_quirks(feats):
    from nerfart_amd import criteria
    x = torch.rand(1, 3, 480, 270)
    assert criteria.CLIPLoss(feats).preprocess(x).shape == (1, 3, 224, 224)
    c = criteria.ContrastiveLoss(feats)
    y = c.preprocess(x)
    assert y.shape == (1, 3, 224, 224)
    # (x+1)/2 on a [0,1] image (the reference's quirk): pixel values in [0.5, 1] before normalisation
    lo = (0.5 - torch.tensor(criteria.CLIP_MEAN)) / torch.tensor(criteria.CLIP_STD)
    assert (y.amin(dim=(0, 2, 3)) >= lo - 0.35).all()
    assert criteria.resize(x, 224).shape[-2:] == (398, 224)                 # shorter side 224, int(224 * 480 / 270)
    p = criteria.PatchNCELoss(feats, (480, 270))
    padded = torch.nn.functional.pad(x, (270, 270, 480, 480))
    assert padded.shape[-2:] == (1440, 810)
    g = torch.Generator().manual_seed(0)
    for (i, j) in p.crop_origins(480, 270, 112, 112, False, generator=g):
        assert 100 <= i < 480 - 112 + 1 - 100 and 0 <= j <= 270 - 112


def test_loss_heads_closed_form(feats):
    from nerfart_amd import criteria
    torch.manual_seed(0)
    gt, pred = torch.rand(1, 3, 96, 64), torch.rand(1, 3, 96, 64, requires_grad=True)
    clip_l = criteria.CLIPLoss(feats)
    v = clip_l(gt, "photo", pred, "painting")
    fs = feats.image_features(clip_l.preprocess(gt)); ft = feats.image_features(clip_l.preprocess(pred))
    e = (ft - fs); e = e / e.norm(dim=-1, keepdim=True)
    d = (feats.text_features("painting") - feats.text_features("photo")).mean(0, keepdim=True); d = d / d.norm()
    np.testing.assert_allclose(float(v), float(1 - (e * d).sum()), rtol=1e-5, atol=1e-6)
    con = criteria.ContrastiveLoss(feats)
    v2 = con(gt, "photo", pred, "painting")
    f = feats.image_features(con.preprocess(pred)); fsrc = feats.image_features(con.preprocess(gt))
    near = (f - feats.text_features("painting") + 1e-6).norm(dim=-1)          # [80]
    far_t = (f - feats.text_features("photo") + 1e-6).norm(dim=-1)
    far_i = (f - fsrc + 1e-6).norm(dim=-1)                                   # [1], broadcast over the 80 templates
    ref = (near ** 2 + torch.clamp(2 - far_t, min=0) ** 2 + torch.clamp(2 - far_i, min=0) ** 2).mean()
    np.testing.assert_allclose(float(v2), float(ref), rtol=1e-5, atol=1e-6)
    pn = criteria.PatchNCELoss(feats, (128, 96), n_patches=2)
    v3 = pn(["photo", "sketch"], pred, "painting", False, crops=[(4, 3), (10, 20)])
    assert v3.ndim == 0 and torch.isfinite(v3)
    # the 12 crops go through the encoder as one batch: same value as the reference's crop-by-crop loop
    xp = criteria.resize(torch.nn.functional.pad(pred, (270, 270, 480, 480)), (128, 96), "bicubic")
    up = lambda t: torch.nn.functional.interpolate(t, size=(224, 224), mode="bicubic", align_corners=False)
    ref3 = sum(pn.patch_loss(["photo", "sketch"], up(xp[..., i:i + 112, j:j + 112]), "painting") for (i, j) in [(4, 3), (10, 20)])
    np.testing.assert_allclose(float(v3), float(ref3), rtol=1e-4, atol=1e-5)
    (v + v2 + v3).backward()
    assert torch.isfinite(pred.grad).all() and float(pred.grad.abs().max()) > 0


def test_create_fine_neg_texts(tmp_path):
    from nerfart_amd import criteria
    p = tmp_path / "neg.txt"
    p.write_text("#portrait\n1.a portrait\n2.a selfie\n#zombie\n1.a zombie\n#other\n1.a photo\n2.a cat\n")
    assert criteria.create_fine_neg_texts("painting, oil on canvas", str(p)) == ["a zombie", "a photo", "a cat"]
    assert criteria.create_fine_neg_texts("a Zombie face", str(p)) == ["a portrait", "a selfie", "a photo", "a cat"]
    assert len(criteria.create_fine_neg_texts("cubism", str(p))) == 5


def test_create_fine_neg_texts_matches_reference_golden(golden):
    """G13: the list the reference's Trainer.create_fine_neg_texts built from its own criteria/neg_text.txt (a data file of the
    reference, read where it lies - build container only)."""
    from nerfart_amd import criteria
    path = "/root/reference/criteria/neg_text.txt"
    if not os.path.exists(path):
        pytest.skip("reference data file not present on this box")
    got = criteria.create_fine_neg_texts("painting, oil on canvas, Vincent van gogh self-portrait style", path)
    assert got == [str(t) for t in golden["G13_neg_texts"]]
    assert criteria.prompt_family("Pixlar") == "disney" and criteria.prompt_family("a sketch of a wolf") == "wolf"
    assert criteria.prompt_family("cubism") is None
