"""The image side of the style losses on the hand-written kernels (rows a20-a22): preprocessing chains as resampling gathers
(`nerfart_resample_fwd/bwd`), the CLIP image encoder (`clip_native`), and the three loss heads with their feature gradients in
one launch (`nerfart_clip_style_heads`) - csrc/style_heads.hip, csrc/clip_vit.hip.

`clip_style_loss(...)` evaluates  w_clip * directional + w_contrastive * global + w_patchnce * local  exactly as
`criteria.CLIPLoss / ContrastiveLoss / PatchNCELoss` restate the reference (clip_loss.py:244-254 on :166-168 preprocessing;
contrastive_loss.py:146-153 on :98-101; patchnce_loss.py:153-215), differentiable w.r.t. the predicted image.  One encoder
batch holds all 4 + P images of a step ("batch the 16 images"); the text features are the cached ones of
criteria.ClipFeatures.
"""
import ctypes as C

import torch

from . import hip

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: {hip.lib.nerfart_last_error().decode()}")


def normalize_affine(device, half_shift: bool):
    """[2, 3]: out = v * a_c + b_c for Normalize(CLIP mean / std), optionally after (x + 1) / 2."""
    mean, std = torch.tensor(CLIP_MEAN), torch.tensor(CLIP_STD)
    a, b = (0.5 / std, (0.5 - mean) / std) if half_shift else (1.0 / std, -mean / std)
    return torch.stack([a, b]).float().contiguous().to(device)


class _Resample(torch.autograd.Function):
    """dst [N, C, Ho, Wo] from src [n_src, C, Hs, Ws]; geo = (pad_t, pad_l, Hp, Wp, Hr, Wr, mode, N, Ho, Wo)."""

    @staticmethod
    def forward(ctx, src, geo, crop, win, affine):
        pad_t, pad_l, Hp, Wp, Hr, Wr, mode, N, Ho, Wo = geo
        x = src.detach().float().contiguous()
        n_src, Cc, Hs, Ws = x.shape
        dst = torch.empty(N, Cc, Ho, Wo, dtype=torch.float32, device=x.device)
        _check(hip.lib.nerfart_resample_fwd(_ptr(x), n_src, Cc, Hs, Ws, pad_t, pad_l, Hp, Wp, Hr, Wr, mode, _ptr(crop), _ptr(win), _ptr(affine),
                                            _ptr(dst), N, Ho, Wo, _stream()), "nerfart_resample_fwd")
        ctx.geo, ctx.shape, ctx.crop, ctx.win, ctx.affine, ctx.dtype = geo, x.shape, crop, win, affine, src.dtype
        return dst

    @staticmethod
    def backward(ctx, g):
        pad_t, pad_l, Hp, Wp, Hr, Wr, mode, N, Ho, Wo = ctx.geo
        n_src, Cc, Hs, Ws = ctx.shape
        g = g.detach().float().contiguous()
        gs = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        _check(hip.lib.nerfart_resample_bwd(_ptr(g), n_src, Cc, Hs, Ws, pad_t, pad_l, Hp, Wp, Hr, Wr, mode, _ptr(ctx.crop), _ptr(ctx.win),
                                            _ptr(ctx.affine), _ptr(gs), N, Ho, Wo, _stream()), "nerfart_resample_bwd")
        return gs.to(ctx.dtype), None, None, None, None


def resample(src, out_hw, resized_hw=None, mode="bicubic", pad=(0, 0, 0, 0), crops=None, windows=None, affine=None):
    """One preprocessing stage.  src [n_src, C, H, W]; pad = (left, right, top, bottom) zeros (F.pad order); the padded canvas (or,
    with `windows` [(y, x, h, w)] per output, that sub-window of the source) is resized to resized_hw (default: out_hw) and output
    n is the out_hw window at crops[n] = (y0, x0).  N outputs = len(crops) / len(windows) / n_src."""
    n_src, _, Hs, Ws = src.shape
    pl, pr, pt, pb = pad
    Hp, Wp = Hs + pt + pb, Ws + pl + pr
    Ho, Wo = out_hw
    Hr, Wr = resized_hw if resized_hw is not None else out_hw
    N = len(crops) if crops is not None else (len(windows) if windows is not None else n_src)
    dev = src.device
    crop_t = torch.tensor(crops, dtype=torch.int32, device=dev).reshape(N, 2) if crops is not None else None
    win_t = torch.tensor(windows, dtype=torch.int32, device=dev).reshape(N, 4) if windows is not None else None
    geo = (pt, pl, Hp, Wp, Hr, Wr, 1 if mode == "bicubic" else 0, N, Ho, Wo)
    return _Resample.apply(src, geo, crop_t, win_t, affine)


class _Heads(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, n_patches, text_dir, t_tgt, t_con, t_neg, weights, margin, tau):
        f = feats.detach().float().contiguous()
        S, T = (t_neg.shape[0], t_neg.shape[1]) if t_neg is not None and t_neg.numel() else (0, t_tgt.shape[0])
        out = torch.empty(4, dtype=torch.float32, device=f.device)
        g = torch.empty_like(f)
        _check(hip.lib.nerfart_clip_style_heads(_ptr(f), n_patches, _ptr(text_dir), _ptr(t_tgt), _ptr(t_con), _ptr(t_neg), S, T,
                                                weights[0], weights[1], weights[2], margin, tau, _ptr(out), _ptr(g), _stream()),
               "nerfart_clip_style_heads")
        ctx.save_for_backward(g)
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        (g,) = ctx.saved_tensors
        return g * g_total, None, None, None, None, None, None, None, None


def _short_side(h, w, size):
    return (size, int(size * w / h)) if h <= w else (int(size * h / w), size)


def clip_style_loss(feats, rgb_pred, rgb_gt, src_text, target_text, con_neg_text, nce_neg_texts, target_hw, crops, is_full_res, weights,
                    margin=2.0, tau=0.07):
    """feats: criteria.ClipFeatures (native).  rgb_pred / rgb_gt [1, 3, H, W] in [0, 1] on the GPU.  crops: the PatchNCE crop
    origins [(i, j)] in the target_hw grid.  Returns (total, parts [4] = total, directional, contrastive, patchnce)."""
    dev = rgb_pred.device
    H, W = rgb_pred.shape[-2:]
    norm, norm_half = normalize_affine(dev, False), normalize_affine(dev, True)
    imgs = []
    # CLIPLoss.preprocess: Resize((224, 224), bicubic), Normalize - prediction, then source (no gradient)
    imgs.append(resample(rgb_pred, (224, 224), mode="bicubic", affine=norm))
    with torch.no_grad():
        imgs.append(resample(rgb_gt, (224, 224), mode="bicubic", affine=norm))
    # ContrastiveLoss.preprocess: (x + 1) / 2, Resize(224) on the shorter side, CenterCrop(224), Normalize
    rh, rw = _short_side(H, W, 224)
    cc = [(int(round((rh - 224) / 2.0)), int(round((rw - 224) / 2.0)))]
    imgs.append(resample(rgb_pred, (224, 224), resized_hw=(rh, rw), mode="bicubic", crops=cc, affine=norm_half))
    with torch.no_grad():
        imgs.append(resample(rgb_gt, (224, 224), resized_hw=(rh, rw), mode="bicubic", crops=cc, affine=norm_half))
    # PatchNCE: ZeroPad2d(270, 270, 480, 480) -> bicubic resize to the dataset's H x W -> crops (112^2 up-sampled x2 bicubic, or
    # 224^2 at full resolution) -> (x + 1) / 2, Resize([224, 224]) (identity at this size), Normalize
    P = len(crops)
    if P:
        Ht, Wt = target_hw
        full = resample(rgb_pred, (Ht, Wt), mode="bicubic", pad=(270, 270, 480, 480))
        th = 224 if is_full_res else 112
        wins = [(i, j, min(th, Ht - i), min(th, Wt - j)) for (i, j) in crops]
        imgs.append(resample(full, (224, 224), mode="bicubic", windows=wins, affine=norm_half))
    batch = torch.cat(imgs, dim=0)
    f = feats.encode_image(batch)
    text_dir = feats.text_direction(src_text, target_text).float().contiguous()
    t_tgt = feats.text_features(target_text).float().contiguous()
    t_con = feats.text_features(con_neg_text).float().contiguous()
    t_neg = torch.stack([feats.text_features(s).float() for s in nce_neg_texts]).contiguous() if P and len(nce_neg_texts) else None
    total, parts = _Heads.apply(f, P, text_dir, t_tgt, t_con, t_neg, tuple(float(w) for w in weights), float(margin), float(tau))
    return total, parts
