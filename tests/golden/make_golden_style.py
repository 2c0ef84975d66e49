"""Golden vectors for the style-loss arithmetic from the REAL reference on CPU:

    python tests/golden/make_golden_style.py          -> tests/golden/style_golden.npz

Runs, where they lie under /root/reference, `criteria/clip_loss.py:CLIPLoss`, `criteria/contrastive_loss.py:ContrastiveLoss`,
`criteria/patchnce_loss.py:PatchNCELoss`, `criteria/perp_loss.py:VGGPerceptualLoss` and `Trainer.calc_style_loss`
(models/frameworks/volsdf.py:878-915) - the heads, the template averaging, the `F.pairwise_distance` broadcast, the ZeroPad2d
"last assignment wins", the crop ranges, the discarded first crop draw and the draw order are all the reference's own code.

What the container lacks is third party and is substituted at the import boundary (none of it is reference arithmetic):
  * `clip` (git+https://github.com/openai/CLIP.git, unpinned; no weights / vocabulary offline): `clip.load` returns this
    repo's random-weight ViT-B/32 (`clip_vit.build_clip("cpu", seed=0)`, fp32; architecture pinned against
    transformers.CLIPModel in tests/test_clip.py) and a preprocess object with the five-entry transform list of clip's
    `_transform`; `clip.tokenize` is the byte-hash stand-in `clip_vit.synthetic_tokens`.  `encode_text` is memoised per token
    batch (the reference re-encodes constants ~100 times per call; results are bit-identical).
  * `torchvision.transforms` (pinned 0.9.1 in the reference's README.md:21): thin tensor versions of Compose / Resize /
    CenterCrop / Normalize / functional.crop with 0.9.1's semantics (F.interpolate, align_corners=False, no antialias; an int
    size matches the SHORTER side, long side = int(size * long / short); centre crop offsets int(round((h - s) / 2.))).
  * `torchvision.models.vgg16`: a torchvision-shaped `features` Sequential (configuration "D") whose first seven convolutions
    carry `vgg.VGG16Features(seed=0)`'s weights (ImageNet weights do not exist offline; the rest is seeded noise - perp_loss.py
    computes the fourth slice and drops it).
The loss classes hard-code `device = "cuda"`; it is set to "cpu" after construction.

Stored per case of tests/style_inputs.py (inputs regenerate from seeds; SHA-256 stored): the three preprocessing outputs
(strided), image features, text direction / features, every loss term, the draws calc_style_loss made (prompts, crop origins),
the total, and d total / d rgb (norm, per-term norms, 4096 sampled entries).
"""
import os
import random
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import make_golden as mg  # noqa: E402
import style_inputs as si  # noqa: E402

CROPS = []          # every transforms.functional.crop call the reference makes: (i, j, h, w)


def install_style_stubs():
    mg.install_stubs()
    from nerfart_amd import clip_vit, vgg
    # ---- clip ------------------------------------------------------------------------------------------------
    model = clip_vit.build_clip("cpu", seed=0)
    raw_encode_text, cache = model.encode_text, {}

    def encode_text(tokens):
        key = tokens.cpu().numpy().tobytes()
        if key not in cache:
            with torch.no_grad():
                cache[key] = raw_encode_text(tokens)
        return cache[key].clone()
    model.encode_text = encode_text

    class InterpolationMode:
        NEAREST, BILINEAR, BICUBIC = "nearest", "bilinear", "bicubic"

    class Compose:
        def __init__(self, transforms):
            self.transforms = list(transforms)

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    class Resize:
        def __init__(self, size, interpolation=InterpolationMode.BILINEAR):
            self.size, self.interpolation = size, interpolation

        def __call__(self, img):            # torchvision 0.9.1 functional_tensor.resize
            size = self.size
            if isinstance(size, int) or len(size) == 1:
                h, w = img.shape[-2:]
                short, long = (w, h) if w <= h else (h, w)
                req = size if isinstance(size, int) else size[0]
                if short == req:
                    return img
                new_short, new_long = req, int(req * long / short)
                new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
            else:
                new_h, new_w = size
            return F.interpolate(img, size=[new_h, new_w], mode=self.interpolation, align_corners=False)

    class CenterCrop:
        def __init__(self, size):
            self.size = (size, size) if isinstance(size, int) else tuple(size)

        def __call__(self, img):
            h, w = img.shape[-2:]
            ch, cw = self.size
            top, left = int(round((h - ch) / 2.)), int(round((w - cw) / 2.))
            return img[..., top:top + ch, left:left + cw]

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, t):
            mean = torch.as_tensor(self.mean, dtype=t.dtype, device=t.device).view(-1, 1, 1)
            std = torch.as_tensor(self.std, dtype=t.dtype, device=t.device).view(-1, 1, 1)
            return (t - mean) / std

    class ToTensor:
        def __call__(self, x):
            raise RuntimeError("PIL path not used")

    def crop(img, top, left, height, width):
        CROPS.append((top, left, height, width))
        return img[..., top:top + height, left:left + width]

    tr = sys.modules["torchvision.transforms"]
    tr.__dict__.update(InterpolationMode=InterpolationMode, Compose=Compose, Resize=Resize, CenterCrop=CenterCrop, Normalize=Normalize,
                       ToTensor=ToTensor)
    tr.functional.crop = crop
    preprocess = Compose([Resize(224, InterpolationMode.BICUBIC), CenterCrop(224), (lambda im: im.convert("RGB")), ToTensor(),
                          Normalize((0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711))])     # clip._transform
    cl = sys.modules["clip"]
    cl.load = lambda name, device="cpu", **kw: (model, preprocess)
    cl.tokenize = lambda strings, **kw: torch.stack([clip_vit.synthetic_tokens(s) for s in ([strings] if isinstance(strings, str) else strings)])

    # ---- torchvision.models.vgg16 ------------------------------------------------------------------------------
    mine = vgg.VGG16Features(seed=0)

    def vgg16(pretrained=False, **kw):
        cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
        layers, cin = [], 3
        gen = torch.Generator().manual_seed(123)
        for v in cfg:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                conv = nn.Conv2d(cin, v, kernel_size=3, padding=1)
                with torch.no_grad():
                    conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * (2.0 / (9 * v)) ** 0.5)
                    conv.bias.zero_()
                layers += [conv, nn.ReLU(inplace=True)]
                cin = v
        feats = nn.Sequential(*layers)
        with torch.no_grad():
            for idx, conv in mine.features.items():
                feats[int(idx)].weight.copy_(conv.weight)
                feats[int(idx)].bias.copy_(conv.bias)
        return types.SimpleNamespace(features=feats)
    sys.modules["torchvision.models"].vgg16 = vgg16
    return model


def strided(t):
    return t[..., ::4, ::4].contiguous()


def main():
    clip_model = install_style_stubs()
    sys.path.insert(0, mg.REF)
    os.chdir(mg.REF)
    from criteria.clip_loss import CLIPLoss
    from criteria.contrastive_loss import ContrastiveLoss
    from criteria.patchnce_loss import PatchNCELoss
    from criteria.perp_loss import VGGPerceptualLoss
    from models.frameworks import volsdf as ref_volsdf
    from nerfart_amd.config import ConfigDict
    torch.set_num_threads(16)
    out = {"clip_state_sha256": np.array(mg.state_checksum(clip_model.state_dict()))}

    for name, (H, W, target_hw, downscale) in si.CASES.items():
        tag = name + "_"
        losses = {"clip": CLIPLoss(), "contrastive": ContrastiveLoss(), "patchnce": PatchNCELoss(list(target_hw)), "perceptual": VGGPerceptualLoss()}
        for k in ("clip", "contrastive", "patchnce"):
            losses[k].device = "cpu"
        args = ConfigDict({"finetune": ConfigDict(dict(src_text=si.SRC_TEXT, target_text=si.TARGET_TEXT, **si.WEIGHTS)),
                           "data": ConfigDict(dict(downscale=downscale))})
        fake = types.SimpleNamespace(loss_dict=losses, neg_texts=None)
        fake.neg_texts = ref_volsdf.Trainer.create_fine_neg_texts(fake, args)
        out[tag + "neg_texts"] = np.array(fake.neg_texts)
        rgb, rgb_gt = si.image_pair(name)
        out[tag + "rgb_sha256"], out[tag + "rgb_gt_sha256"] = np.array(si.sha(rgb)), np.array(si.sha(rgb_gt))
        img = lambda t: t.reshape(1, H, W, 3).permute(0, 3, 1, 2)

        # ---- preprocessing chains + features, head by head (the reference's own preprocess objects) ----
        with torch.no_grad():
            out[tag + "pre_clip"] = strided(losses["clip"].preprocess(img(rgb)))
            out[tag + "pre_contrastive"] = strided(losses["contrastive"].preprocess(img(rgb)))
            out[tag + "feat_clip_pred"] = losses["clip"].get_image_features(img(rgb))
            out[tag + "feat_clip_gt"] = losses["clip"].get_image_features(img(rgb_gt))
            out[tag + "feat_contrastive_pred"] = losses["contrastive"].get_image_features(img(rgb))
            out[tag + "text_direction"] = losses["clip"].compute_text_direction(si.SRC_TEXT, si.TARGET_TEXT)
            out[tag + "text_target"] = losses["clip"].get_text_features(si.TARGET_TEXT)
            padded = losses["patchnce"].resize(losses["patchnce"].ZeroPad(img(rgb)))
            out[tag + "patchnce_canvas"] = strided(padded)

        # ---- calc_style_loss: the whole objective with the reference's own draws ----
        x = rgb.clone().requires_grad_(True)
        random.seed(si.DRAW_SEED)
        torch.manual_seed(si.DRAW_SEED)
        del CROPS[:]
        choice_log, sample_log = [], []
        orig_choice, orig_sample = random.choice, random.sample
        random.choice = lambda seq: (choice_log.append(orig_choice(seq)) or choice_log[-1])
        random.sample = lambda pop, k: (sample_log.append(orig_sample(pop, k)) or sample_log[-1])
        parts = {}
        hooks = []
        for k, m in losses.items():
            hooks.append(m.register_forward_hook(lambda mod, inp, res, k=k: parts.__setitem__(k, res)))
        try:
            total = ref_volsdf.Trainer.calc_style_loss(fake, x, rgb_gt, args, H)
        finally:
            random.choice, random.sample = orig_choice, orig_sample
            for h in hooks:
                h.remove()
        assert len(choice_log) == 1 and len(sample_log) == 1 and len(CROPS) == 12, (choice_log, sample_log, CROPS)
        out[tag + "draw_contrastive_text"] = np.array(choice_log[0])
        out[tag + "draw_patchnce_texts"] = np.array(sample_log[0])
        out[tag + "draw_crops"] = np.array(CROPS, dtype=np.int64)
        for k, v in parts.items():
            out[tag + "loss_" + k] = v.detach()
        out[tag + "loss_total"] = total.detach()
        idx = si.grad_sample_index(name)
        grads = {}
        for k, v in list(parts.items()) + [("total", total)]:
            (g,) = torch.autograd.grad(v, x, retain_graph=True)
            grads[k] = g
            out[tag + "gradnorm_" + k] = g.norm()
            out[tag + "gradsample_" + k] = g.reshape(-1)[idx].clone()
        print(f"{name}: total {float(total):.6f}  " + "  ".join(f"{k} {float(v):.6f} (|g| {float(grads[k].norm()):.3e})" for k, v in parts.items()),
              " contrastive negative:", choice_log[0], " crops:", CROPS[:3], "...")

        # ---- the PatchNCE crops after the reference's x2 up-sampling + preprocess (first two crops, strided) ----
        with torch.no_grad():
            for n, (i, j, th, tw) in enumerate(CROPS[:2]):
                c = padded[..., i:i + th, j:j + tw]
                if downscale != 1:
                    c = F.interpolate(c, size=(224, 224), mode="bicubic", align_corners=False)
                out[tag + f"pre_patchnce_{n}"] = strided(losses["patchnce"].preprocess(c))

    np.savez_compressed(os.path.join(HERE, "style_golden.npz"), **mg.t2n(out))
    print("wrote style_golden.npz", os.path.getsize(os.path.join(HERE, "style_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
