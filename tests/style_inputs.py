"""Seeded inputs shared by tests/golden/make_golden_style.py (build container, runs the reference) and the tests that hold the
package to those vectors (tests/test_style_golden.py, tests/test_gpu_style_golden.py).  Only torch.rand + exact IEEE arithmetic
(no interpolation, no transcendental functions), so the images regenerate bit-identically on any box; the fixture stores a
SHA-256 of each to detect RNG drift."""
import hashlib

import torch

# name -> (H, W, target_hw of PatchNCELoss, data.downscale).  `cfg3` is BASELINE.json configs[2] (480 x 270, downscale 2);
# `square` exercises the H == W crop branch (patchnce_loss.py:203-208); `full` the is_full_res branch (224^2 crops, rows 200..).
CASES = {
    "cfg3": (480, 270, (480, 270), 2),
    "square": (256, 256, (256, 256), 2),
    "full": (960, 540, (960, 540), 1),
}
SRC_TEXT = "photo"
TARGET_TEXT = "painting, oil on canvas, Vincent van gogh self-portrait style"      # configs/volsdf_fangzhou_vangogh.yaml:81
WEIGHTS = dict(w_clip=1.0, w_perceptual=2.0, w_contrastive=0.2, w_patchnce=0.1)    # configs/volsdf_fangzhou_vangogh.yaml:82-86
DRAW_SEED = 5            # random.seed / torch.manual_seed before calc_style_loss
N_GRAD_SAMPLES = 4096


def image_pair(name):
    """(rgb_pred, rgb_gt) [1, H*W, 3] in [0, 1] as the renderer hands them to calc_style_loss: 8 x 8 blocks + pixel noise."""
    H, W, _, _ = CASES[name]
    g = torch.Generator().manual_seed(1000 + sorted(CASES).index(name))
    out = []
    for _ in range(2):
        coarse = torch.rand(1, 3, (H + 7) // 8, (W + 7) // 8, generator=g)
        img = coarse.repeat_interleave(8, dim=2).repeat_interleave(8, dim=3)[:, :, :H, :W] * 0.75 + torch.rand(1, 3, H, W, generator=g) * 0.25
        out.append(img.permute(0, 2, 3, 1).reshape(1, H * W, 3).contiguous())
    return out[0], out[1]


def sha(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


def grad_sample_index(name):
    H, W, _, _ = CASES[name]
    g = torch.Generator().manual_seed(77)
    return torch.randperm(H * W * 3, generator=g)[:N_GRAD_SAMPLES]
