"""The hand-written CLIP ViT-B/32 image encoder (rows a23 / B4; csrc/clip_vit.hip through the C ABI) against the oracle:
`clip_vit.CLIP` in fp32 on the CPU with the same random weights (its architecture is pinned against transformers.CLIPModel
in tests/test_clip.py; parity against the OpenAI weights is UNPINNED - none on disk).

Tolerances: the kernels compute with fp16 weights and fp16 GEMM operands (as clip.load(device="cuda") does), fp32
accumulation / LayerNorm / softmax: features agree with the fp32 oracle (which uses the fp16-ROUNDED weights, so only the
activation rounding differs) to ~1e-3 of their scale; pixel gradients to ~1 %.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_gemm_kernel_matches_matmul():
    from nerfart_amd import clip_native
    g = torch.Generator().manual_seed(0)
    for (M, N, K) in ((64, 64, 64), (128, 192, 256), (832, 2304, 768), (64, 512, 768), (832, 768, 3072)):
        a = torch.randn(M, K, generator=g).half()
        w = (torch.randn(N, K, generator=g) * 0.05).half()          # asymmetric operands: a transposed store cannot pass
        c = clip_native.gemm_f16_nt(a.to(DEV), w.to(DEV)).cpu()
        ref = a.double() @ w.double().t()
        err = (c.double() - ref).abs().max().item()
        print(f"  gemm {M}x{N}x{K}: max abs err {err:.2e} (ref max {ref.abs().max().item():.2f})")
        assert err < 2e-3 * max(1.0, ref.abs().max().item())


def test_gemm_kernel_with_the_weight_read_in_place_matches_matmul():
    """C = A . Wt with Wt [K, N] row-major (the backward GEMMs on the forward weight matrices: B fragments through the
    transposing LDS read); the result must also equal, bit for bit, the NT kernel on the explicitly transposed operand - same
    products, same accumulation order."""
    from nerfart_amd import clip_native
    g = torch.Generator().manual_seed(1)
    for (M, N, K) in ((64, 64, 64), (128, 192, 256), (832, 768, 2304), (64, 768, 512), (832, 3072, 768)):
        a = torch.randn(M, K, generator=g).half()
        wt = (torch.randn(K, N, generator=g) * 0.05).half()
        wt[:, 1::2] *= 0.5                                           # columns differ in scale: a permuted column cannot pass
        c = clip_native.gemm_f16_nn(a.to(DEV), wt.to(DEV))
        ref = a.double() @ wt.double()
        err = (c.cpu().double() - ref).abs().max().item()
        print(f"  gemm nn {M}x{N}x{K}: max abs err {err:.2e} (ref max {ref.abs().max().item():.2f})")
        assert err < 2e-3 * max(1.0, ref.abs().max().item())
        assert torch.equal(c, clip_native.gemm_f16_nt(a.to(DEV), wt.t().contiguous().to(DEV)))


def _models():
    from nerfart_amd import clip_vit
    gpu = clip_vit.build_clip(DEV, seed=0)                           # fp16 weights, as clip.load on a GPU
    cpu = clip_vit.build_clip("cpu", seed=0)
    with torch.no_grad():                                            # the oracle runs fp32 arithmetic on the SAME (fp16-rounded) weights
        for (n, p), (_, q) in zip(cpu.named_parameters(), gpu.named_parameters()):
            p.copy_(q.detach().float().cpu())
    return gpu, cpu


@pytest.mark.parametrize("B", [1, 3, 16])
def test_image_features_match_fp32_oracle(B):
    from nerfart_amd import clip_native
    gpu, cpu = _models()
    enc = clip_native.NativeImageEncoder(gpu)
    g = torch.Generator().manual_seed(B)
    img = torch.randn(B, 3, 224, 224, generator=g)
    with torch.no_grad():
        ref = cpu.encode_image(img)
        got = enc(img.to(DEV)).cpu()
        lib = gpu.encode_image(img.to(DEV)).float().cpu()            # the torch / library path in fp16, for scale
    scale = ref.abs().max().item()
    err, err_lib = (got - ref).abs().max().item(), (lib - ref).abs().max().item()
    print(f"  B={B}: native max err {err:.3e}, torch-fp16 module max err {err_lib:.3e}, feature scale {scale:.3f}")
    assert got.shape == (B, 512) and torch.isfinite(got).all()
    assert err < 1e-3 * scale                                        # measured 3.3e-4 .. 3.5e-4 of the scale (round 2)
    cos = torch.nn.functional.cosine_similarity(got, ref, dim=-1)
    assert cos.min() > 0.9999


def test_pixel_gradient_matches_fp32_autograd():
    from nerfart_amd import clip_native
    gpu, cpu = _models()
    enc = clip_native.NativeImageEncoder(gpu)
    g = torch.Generator().manual_seed(7)
    B = 3
    img = torch.randn(B, 3, 224, 224, generator=g)
    cot = torch.randn(B, 512, generator=g) * 1e-3                    # small cotangent: exercises the fp16 loss-scale path
    x = img.clone().requires_grad_(True)
    (cpu.encode_image(x) * cot).sum().backward()
    xg = img.to(DEV).requires_grad_(True)
    f = enc(xg)
    (f * cot.to(DEV)).sum().backward()
    ref, got = x.grad, xg.grad.cpu()
    rel = ((got - ref).norm() / ref.norm()).item()
    cos = torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0).item()
    print(f"  d(feat . c)/d img: rel err {rel:.3e}, cosine {cos:.6f}, |ref| {ref.norm().item():.3e}")
    assert torch.isfinite(got).all() and rel < 5e-3 and cos > 0.9999      # measured 8.1e-4
    # linearity in the cotangent (the device-side power-of-two scale must cancel exactly)
    xg2 = img.to(DEV).requires_grad_(True)
    (enc(xg2) * (1024.0 * cot.to(DEV))).sum().backward()
    np.testing.assert_allclose((xg2.grad / 1024.0).cpu().numpy(), got.numpy(), rtol=0, atol=1e-6 * got.abs().max().item())
    # no gradient requested: the forward keeps nothing and agrees with the grad-enabled forward
    with torch.no_grad():
        f0 = enc(img.to(DEV))
    assert torch.equal(f0, f.detach())


def test_style_heads_run_on_the_native_encoder():
    """criteria.ClipFeatures on a GPU model uses the native encoder by default; the three heads' values agree with the torch
    module path and the pixel gradient flows."""
    from nerfart_amd import criteria, clip_vit
    g = torch.Generator().manual_seed(2)
    gt, pred = torch.rand(1, 3, 120, 68, generator=g).to(DEV), torch.rand(1, 3, 120, 68, generator=g).to(DEV)
    out = {}
    for native in (True, False):
        feats = criteria.ClipFeatures(model=clip_vit.build_clip(DEV, seed=0), device=DEV, synthetic=True, native=native,
                                      templates=["a photo of a {}.", "a sketch of a {}.", "art of the {}."])
        assert feats.native is native
        p = pred.clone().requires_grad_(True)
        l1 = criteria.CLIPLoss(feats)(gt, "photo", p, "painting")
        l2 = criteria.ContrastiveLoss(feats)(gt, "photo", p, "painting")
        l3 = criteria.PatchNCELoss(feats, (120, 68), n_patches=2)(["photo", "sketch"], p, "painting", False, crops=[(3, 2), (9, 11)])
        (l1 + l2 + l3).float().backward()
        out[native] = ([float(l1), float(l2), float(l3)], p.grad.detach().float().cpu())
    for a, b in zip(out[True][0], out[False][0]):
        assert abs(a - b) <= 2e-2 * max(1.0, abs(b)), out
    cos = torch.nn.functional.cosine_similarity(out[True][1].flatten(), out[False][1].flatten(), dim=0).item()
    print(f"  heads native {out[True][0]} torch {out[False][0]}; pixel-gradient cosine {cos:.5f}")
    assert cos > 0.99
