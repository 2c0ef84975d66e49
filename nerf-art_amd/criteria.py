"""CLIP-driven style losses of the fine-tune step (rows a20-a22): directional CLIP loss, global contrastive loss,
local PatchNCE loss (+ the VGG perceptual term of vgg.py) - the arithmetic of criteria/clip_loss.py:155-304, contrastive_loss.py:91-186 and
patchnce_loss.py:91-220, including the quirks SURVEY.md (a21, a22, Appendix C) lists.

One CLIP instance is shared by the three heads (the reference loads three identical copies) and text features are
cached per class string (the reference re-encodes 80 templates x up to 11 prompts on every call; they are
constants).  Text enters as token ids through `tokenize` - with the real BPE vocabulary pass `clip.tokenize`;
`clip_vit.synthetic_tokens` is the offline stand-in (random-weight benchmarks).  `templates`: the reference's
80 ImageNet prompt templates are data (criteria/clip_loss.py:9-90); pass them for checkpoint parity - the default
generates 80 distinct templates so that every broadcast has the reference's shapes.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import clip_vit

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
DEFAULT_TEMPLATES = ["a photo of a {}."] + [f"a rendition {i} of a {{}}." for i in range(1, 80)]


def _normalize(x):
    mean = torch.tensor(CLIP_MEAN, device=x.device, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD, device=x.device, dtype=x.dtype).view(1, 3, 1, 1)
    return (x - mean) / std


def resize(x, size, mode="bicubic"):
    """torchvision 0.9 tensor Resize: F.interpolate, align_corners=False, no antialias.  size: (h, w) or the
    shorter side (aspect kept, long side = int(size * long / short))."""
    if isinstance(size, int):
        h, w = x.shape[-2:]
        size = (size, int(size * w / h)) if h <= w else (int(size * h / w), size)
    return F.interpolate(x, size=tuple(size), mode=mode, align_corners=False)


def center_crop(x, s):
    h, w = x.shape[-2:]
    top, left = int(round((h - s) / 2.0)), int(round((w - s) / 2.0))
    return x[..., top:top + s, left:left + s]


class ClipFeatures(nn.Module):
    """Shared encoder + cached, normalised text features [n_templates, 512] per class string."""

    def __init__(self, model: clip_vit.CLIP = None, tokenize=None, templates=None, device="cuda"):
        super().__init__()
        self.model = model if model is not None else clip_vit.build_clip(device)
        self.tokenize = tokenize or (lambda strings: torch.stack([clip_vit.synthetic_tokens(s) for s in strings]))
        self.templates = list(templates) if templates is not None else DEFAULT_TEMPLATES
        self._text = {}

    @property
    def device(self):
        return self.model.positional_embedding.device

    def image_features(self, images, norm=True):
        f = self.model.encode_image(images.to(self.device))
        return f / f.clone().norm(dim=-1, keepdim=True) if norm else f

    @torch.no_grad()
    def text_features(self, class_str: str, norm=True):
        key = (class_str, norm)
        if key not in self._text:
            tokens = self.tokenize([t.format(class_str) for t in self.templates]).to(self.device)
            f = self.model.encode_text(tokens).detach()
            self._text[key] = f / f.norm(dim=-1, keepdim=True) if norm else f
        return self._text[key]


class CLIPLoss(nn.Module):
    """Directional loss: 1 - cos(normalize(E(pred) - E(gt)), normalize(mean_t(E(tgt_t) - E(src_t))))
    (clip_loss.py:234-254, :297).  Images [B,3,H,W] in [0,1]; preprocess = Resize((224,224), bicubic) + normalise."""

    def __init__(self, feats: ClipFeatures):
        super().__init__()
        self.feats = feats
        self.text_direction = None

    def preprocess(self, x):
        return _normalize(resize(x, (224, 224), "bicubic"))

    def compute_text_direction(self, source_class, target_class):
        d = (self.feats.text_features(target_class) - self.feats.text_features(source_class)).mean(dim=0, keepdim=True)
        return d / d.norm(dim=-1, keepdim=True)

    def forward(self, src_img, source_class, target_img, target_class):
        if self.text_direction is None:                     # cached on first use like clip_loss.py:245-246
            self.text_direction = self.compute_text_direction(source_class, target_class)
        src = self.feats.image_features(self.preprocess(src_img))
        tgt = self.feats.image_features(self.preprocess(target_img))
        edit = tgt - src
        edit = edit / edit.clone().norm(dim=-1, keepdim=True)
        return (1.0 - F.cosine_similarity(edit, self.text_direction)).mean()


class ContrastiveLoss(nn.Module):
    """Global contrastive loss (contrastive_loss.py:140-153, euclidean): mean(d(f, T_tgt)^2 + relu(m - d(f, T_src))^2
    + relu(m - d(f, f_src))^2), pairwise_distance broadcasting [1,512] against [80,512].  Preprocess: (x+1)/2 (applied
    to the [0,1] render as the reference does), Resize(224, bicubic) on the shorter side, CenterCrop(224), normalise."""

    def __init__(self, feats: ClipFeatures, margin=2.0):
        super().__init__()
        self.feats, self.margin = feats, margin

    def preprocess(self, x):
        return _normalize(center_crop(resize((x + 1.0) / 2.0, 224, "bicubic"), 224))

    def forward(self, src_img, source_class, target_img, target_class):
        s_txt = self.feats.text_features(source_class)
        t_txt = self.feats.text_features(target_class)
        src = self.feats.image_features(self.preprocess(src_img))
        tgt = self.feats.image_features(self.preprocess(target_img))
        near = F.pairwise_distance(tgt, t_txt.detach(), keepdim=True)
        far_text = F.pairwise_distance(tgt, s_txt.detach(), keepdim=True)
        far_img = F.pairwise_distance(tgt, src.detach(), keepdim=True)
        return torch.mean(near ** 2 + torch.clamp(self.margin - far_text, min=0.0) ** 2
                          + torch.clamp(self.margin - far_img, min=0.0) ** 2)


class PatchNCELoss(nn.Module):
    """Local contrastive loss (patchnce_loss.py:153-220): zero-pad (270,270,480,480), bicubic resize to the dataset's
    H x W, 12 random crops (112^2 up-sampled x2 bicubic unless full resolution), each: (x+1)/2, Resize([224,224])
    (bilinear), normalise, -log(e^{cos+/tau} / (e^{cos+/tau} + sum_neg e^{cos-/tau})), tau = 0.07, cos broadcast over
    the templates; the 12 losses are summed.  `crops`: optional list of 12 (i, j) for deterministic runs."""

    def __init__(self, feats: ClipFeatures, target_hw, n_patches=12):
        super().__init__()
        self.feats, self.target_hw, self.n_patches = feats, tuple(target_hw), n_patches
        self.temperature = 0.07

    def preprocess(self, x):
        return _normalize(resize((x + 1.0) / 2.0, (224, 224), "bilinear"))

    def patch_loss(self, source_classes, img, target_class):
        tgt_txt = self.feats.text_features(target_class)
        f = self.feats.image_features(self.preprocess(img))
        pos = torch.exp(F.cosine_similarity(f, tgt_txt.detach()) / self.temperature)
        neg = 0
        for s in source_classes:
            neg = neg + torch.exp(F.cosine_similarity(f, self.feats.text_features(s).detach()) / self.temperature)
        return torch.mean(-torch.log(pos / (pos + neg)))

    def crop_origins(self, H, W, th, tw, is_full_res, generator=None):
        out = []
        for _ in range(self.n_patches):
            m = (200 if is_full_res else 100) if H != W else (80 if is_full_res else 40)
            i = torch.randint(m, H - th + 1 - m, size=(1,), generator=generator).item()
            j = torch.randint(0, W - tw + 1, size=(1,), generator=generator).item()
            out.append((i, j))
        return out

    def forward(self, source_classes, target_img, target_class, is_full_res: bool, crops=None):
        x = F.pad(target_img, (270, 270, 480, 480))
        x = resize(x, self.target_hw, "bicubic")
        H, W = x.shape[-2:]
        th = tw = 224 if is_full_res else 112
        crops = crops if crops is not None else self.crop_origins(H, W, th, tw, is_full_res)
        imgs = []
        for (i, j) in crops:
            img = x[..., i:i + th, j:j + tw]
            imgs.append(img if is_full_res else F.interpolate(img, size=(224, 224), mode="bicubic", align_corners=False))
        if x.shape[0] == 1 and all(im.shape == imgs[0].shape for im in imgs):
            # the 12 crops as ONE batch through the image encoder (each crop's loss depends on its own features only;
            # the reference's 12 B=1 encodes are launch-bound)
            f = self.feats.image_features(self.preprocess(torch.cat(imgs, dim=0)))[:, None, :]      # [12, 1, 512]
            pos = torch.exp(F.cosine_similarity(f, self.feats.text_features(target_class).detach()[None], dim=-1) / self.temperature)
            neg = 0
            for s in source_classes:
                neg = neg + torch.exp(F.cosine_similarity(f, self.feats.text_features(s).detach()[None], dim=-1) / self.temperature)
            return (-torch.log(pos / (pos + neg))).mean(dim=1).sum()
        total = 0
        for img in imgs:
            total = total + self.patch_loss(source_classes, img, target_class)
        return total


def create_fine_neg_texts(target_text: str, path: str = "criteria/neg_text.txt"):
    """Negative prompts of the reference Trainer (volsdf.py:649-681): `path` is the reference's data file
    (sections "#key", lines "<n>.<text>"); the section that matches the target prompt's family is left out
    (portrait / zombie / wolf / disney / sketch), every other section's texts are concatenated in file order."""
    results, key = {}, 0
    with open(path, "r") as fr:
        for item in fr.readlines():
            item = item.strip()
            if item.startswith("#"):
                key = item[1:]
                results[key] = []
            else:
                results[key].append(item.split(".")[1])
    t = target_text.lower()
    remove = []
    if any(w in t for w in ("botero", "monalisa", "portrait", "painting")):
        remove = ["portrait"]
    elif "zombie" in t:
        remove = ["zombie"]
    elif "wolf" in t:
        remove = ["wolf"]
    elif "pixlar" in t or "disney" in t:
        remove = ["disney"]
    elif "sketch" in t:
        remove = ["sketch"]
    out = []
    for k in results:
        if k not in remove:
            out += results[k]
    return out


class StyleLoss(nn.Module):
    """calc_style_loss (volsdf.py:878-915): w_clip * directional + w_perceptual * VGG + w_contrastive * global +
    w_patchnce * local.  `perceptual`: a `vgg.VGGPerceptualLoss` (pass torchvision's vgg16 weights to it
    for the reference's objective); None leaves the term out."""

    def __init__(self, feats: ClipFeatures, target_hw, src_text="photo", target_text="painting", neg_texts=("photo",),
                 w_clip=1.0, w_contrastive=0.2, w_patchnce=0.1, is_full_res=False, seed=0, perceptual=None, w_perceptual=2.0):
        super().__init__()
        self.clip, self.contrastive = CLIPLoss(feats), ContrastiveLoss(feats)
        self.patchnce = PatchNCELoss(feats, target_hw)
        self.perceptual, self.w_perceptual = perceptual, w_perceptual
        self.src_text, self.target_text, self.neg_texts = src_text, target_text, list(neg_texts)
        self.w = (w_clip, w_contrastive, w_patchnce)
        self.is_full_res = is_full_res
        self.gen = torch.Generator().manual_seed(seed)

    def forward(self, rgb_pred, rgb_gt):
        loss = self.w[0] * self.clip(rgb_gt, self.src_text, rgb_pred, self.target_text)
        if self.perceptual is not None:
            loss = loss + self.w_perceptual * self.perceptual(rgb_pred, rgb_gt)
        k = torch.randint(0, len(self.neg_texts), (1,), generator=self.gen).item()
        loss = loss + self.w[1] * self.contrastive(rgb_gt, self.neg_texts[k], rgb_pred, self.target_text)
        idx = torch.randperm(len(self.neg_texts), generator=self.gen)[:8].tolist()
        H, W = self.patchnce.target_hw
        th = 224 if self.is_full_res else 112
        crops = self.patchnce.crop_origins(H, W, th, th, self.is_full_res, generator=self.gen)
        loss = loss + self.w[2] * self.patchnce([self.neg_texts[i] for i in idx], rgb_pred, self.target_text, self.is_full_res, crops=crops)
        return loss.float()
