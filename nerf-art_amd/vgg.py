"""VGG16 perceptual loss of the fine-tune objective (reference criteria/perp_loss.py:9-57, weight
`w_perceptual` = 2.0 in configs/volsdf_fangzhou_vangogh.yaml:82, used at volsdf.py:898-899).

The reference runs torchvision's `vgg16(pretrained=True).features` in four slices ([:4], [4:9], [9:16], [16:23]) on the
ImageNet-normalised, bilinearly 224 x 224-resized prediction and target and takes the L1 distance of the THIRD slice's
output only (relu3_3; `if i == 2`, perp_loss.py:51) - the fourth slice is computed and dropped, so it is not built here.
The seven 3 x 3 convolutions are im2col (`F.unfold`) + one library GEMM each: the GEMM formulation never goes through
MIOpen, whose first use of a new convolution shape compiles kernels for minutes on a fresh GPU box.  Prediction and
target go through the net as one batch of two.

Weights: pass torchvision's `vgg16` state dict (`features.N.weight / bias`; N = 0, 2, 5, 7, 10, 12, 14 are read).  No
ImageNet checkpoint exists offline, so PARITY IS UNPINNED against the pretrained network; the default is torchvision's
own initialiser (seeded).  The arithmetic is pinned against `F.conv2d` / `nn.Sequential` (tests/test_vgg.py).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

# torchvision vgg16 "D" configuration up to relu3_3: index in `features` -> (in, out); "M" = MaxPool2d(2, 2) before it
_CONVS = [(0, 3, 64, False), (2, 64, 64, False), (5, 64, 128, True), (7, 128, 128, False), (10, 128, 256, True), (12, 256, 256, False),
          (14, 256, 256, False)]
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def conv3x3_gemm(x, weight, bias):
    """3 x 3, stride 1, padding 1 convolution as im2col + GEMM: [B, C, H, W] -> [B, O, H, W]."""
    B, C, H, W = x.shape
    cols = F.unfold(x, kernel_size=3, padding=1)                          # [B, C * 9, H * W]
    out = torch.matmul(weight.reshape(weight.shape[0], -1), cols)         # [B, O, H * W]
    return (out + bias[None, :, None]).reshape(B, weight.shape[0], H, W)


class VGG16Features(nn.Module):
    """features[:16] of torchvision's vgg16 (through relu3_3) with torchvision's parameter names."""

    def __init__(self, state_dict=None, seed: int = 0):
        super().__init__()
        self.features = nn.ModuleDict()
        g = torch.random.get_rng_state()
        torch.manual_seed(seed)
        for idx, cin, cout, _ in _CONVS:
            conv = nn.Conv2d(cin, cout, kernel_size=3, padding=1)
            nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")      # torchvision's VGG._initialize_weights
            nn.init.constant_(conv.bias, 0)
            self.features[str(idx)] = conv
        torch.random.set_rng_state(g)
        if state_dict is not None:
            self.load_state_dict({k: v for k, v in state_dict.items() if k.startswith("features.") and k.split(".")[1] in self.features})
        self.requires_grad_(False)

    def forward(self, x):
        for idx, _, _, pool_first in _CONVS:
            if pool_first:
                x = F.max_pool2d(x, kernel_size=2, stride=2)
            conv = self.features[str(idx)]
            x = F.relu(conv3x3_gemm(x, conv.weight, conv.bias))
        return x


class VGGPerceptualLoss(nn.Module):
    """L1(relu3_3(pred), relu3_3(target)) on ImageNet-normalised, 224 x 224 bilinear inputs (perp_loss.py:27-55)."""

    def __init__(self, state_dict=None, resize: bool = True, seed: int = 0):
        super().__init__()
        self.net = VGG16Features(state_dict, seed)
        self.register_buffer("mean", torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor(IMAGENET_STD).view(1, 3, 1, 1))
        self.resize = resize

    def forward(self, input, target):
        if input.shape[1] != 3:
            input, target = input.repeat(1, 3, 1, 1), target.repeat(1, 3, 1, 1)
        xy = torch.cat([input, target.to(input.dtype)], dim=0)
        xy = (xy - self.mean) / self.std
        if self.resize:
            xy = F.interpolate(xy, mode="bilinear", size=(224, 224), align_corners=False)
        f = self.net(xy)
        n = input.shape[0]
        return F.l1_loss(f[:n], f[n:])
