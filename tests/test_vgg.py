"""VGG16 perceptual loss, CPU.  PARITY UNPINNED against the ImageNet weights (no checkpoint offline);
the arithmetic is pinned against torch's own convolution and a torchvision-shaped nn.Sequential."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _tv_features():
    """torchvision.models.vgg16().features[:23] re-stated from its published configuration 'D'."""
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512]
    layers, cin = [], 3
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return nn.Sequential(*layers)


def test_conv_gemm_is_conv2d():
    from nerfart_amd.vgg import conv3x3_gemm
    g = torch.Generator().manual_seed(0)
    x, w, b = torch.randn(2, 5, 9, 7, generator=g), torch.randn(4, 5, 3, 3, generator=g), torch.randn(4, generator=g)
    np.testing.assert_allclose(conv3x3_gemm(x, w, b).numpy(), F.conv2d(x, w, b, padding=1).numpy(), atol=2e-5, rtol=1e-5)


def test_loss_matches_the_reference_formulation():
    """perp_loss.py:27-55 on a torchvision-shaped net: four slices run, L1 of the third slice's output."""
    from nerfart_amd.vgg import VGGPerceptualLoss, IMAGENET_MEAN, IMAGENET_STD
    torch.manual_seed(3)
    feats = _tv_features().eval()
    sd = {"features." + k: v for k, v in feats.state_dict().items()}
    sd["classifier.0.weight"] = torch.zeros(1)                                   # keys outside features[:16] are ignored
    mine = VGGPerceptualLoss(state_dict=sd)
    assert not any(p.requires_grad for p in mine.parameters())
    pred = torch.rand(1, 3, 48, 27, requires_grad=True)
    gt = torch.rand(1, 3, 48, 27)
    loss = mine(pred, gt)
    mean, std = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1), torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    x = F.interpolate((pred - mean) / std, mode="bilinear", size=(224, 224), align_corners=False)
    y = F.interpolate((gt - mean) / std, mode="bilinear", size=(224, 224), align_corners=False)
    ref = 0.0
    for i, sl in enumerate((feats[:4], feats[4:9], feats[9:16], feats[16:23])):
        x, y = sl(x), sl(y)
        if i == 2:
            ref = ref + F.l1_loss(x, y)
    np.testing.assert_allclose(float(loss), float(ref), rtol=2e-5)
    loss.backward()
    g1 = pred.grad.clone()
    pred.grad = None
    ref.backward()
    np.testing.assert_allclose(g1.numpy(), pred.grad.numpy(), atol=1e-7, rtol=2e-3)
    # single-channel inputs are repeated (perp_loss.py:28-30)
    assert torch.isfinite(mine(torch.rand(1, 1, 20, 20), torch.rand(1, 1, 20, 20)))


import pytest  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("hw", [(60, 34), (480, 270)])
def test_vgg_loss_on_the_gpu_matches_cpu(hw):
    """The hand-written path (resample gather -> implicit-GEMM conv stack on the fp32 matrix-core instructions -> L1 -> backward
    to the pixels; csrc/vgg_conv.hip) against the fp32 torch formulation on the CPU with the same weights.  The gradient is
    piecewise constant in the ReLU / max-pool / sign decisions (a few sit within round-off: tests/test_style_golden.py), so it is
    compared as a whole."""
    from nerfart_amd.vgg import VGGPerceptualLoss
    mine = VGGPerceptualLoss(seed=1)
    g = torch.Generator().manual_seed(2)
    pred, gt = torch.rand(1, 3, *hw, generator=g), torch.rand(1, 3, *hw, generator=g)
    p_cpu = pred.clone().requires_grad_(True)
    l_cpu = mine(p_cpu, gt)
    l_cpu.backward()
    dev = mine.to("cuda")
    p_gpu = pred.cuda().requires_grad_(True)
    l_gpu = dev(p_gpu, gt.cuda())
    (3.0 * l_gpu).backward()                                   # an upstream factor, as StyleLoss applies w_perceptual
    rel = float((p_gpu.grad.cpu() / 3.0 - p_cpu.grad).norm() / p_cpu.grad.norm())
    cos = float(torch.nn.functional.cosine_similarity(p_gpu.grad.cpu().flatten(), p_cpu.grad.flatten(), dim=0))
    print(f"  vgg {hw}: loss gpu {float(l_gpu):.6f} cpu {float(l_cpu):.6f}; pixel gradient rel err {rel:.3e}, cosine {cos:.5f}")
    np.testing.assert_allclose(float(l_gpu), float(l_cpu), rtol=1e-4)
    # round 2 (fp16 operands): 60 x 34: 2.0e-2 / 0.99979; 480 x 270: 6.3e-2 / 0.99800.  fp32 operands (round 3): see profiles/r03*
    assert rel < 2e-2 and cos > 0.9998, (rel, cos)
    # the torch formulation on the GPU (native switched off) agrees as well
    dev.native = False
    l_t = dev(pred.cuda(), gt.cuda())
    np.testing.assert_allclose(float(l_t), float(l_cpu), rtol=1e-3)
    # no gradient requested: nothing is kept, same value
    dev.native = True
    with torch.no_grad():
        assert abs(float(dev(pred.cuda(), gt.cuda())) - float(l_gpu)) < 1e-6
