"""INTEGRATION.md section B is the binding a reference maintainer would paste into `models/frameworks/_nerfart_hip.py`: this test
executes that very code block (only the library path is pointed at the in-tree build) and holds its two functions to the package's
own wrappers - the documented argument lists cannot drift from include/nerfart_hip.h unnoticed."""
import os
import re

import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _doc_binding():
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    code = next(b for b in blocks if "_nerfart_hip.py" in b and "def forward_surface" in b)
    from nerfart_amd import hip
    code = code.replace('C.CDLL("libnerfart_hip.so")', f'C.CDLL({hip.LIB_PATH!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md#B", "exec"), ns)
    return ns


def test_documented_ctypes_binding_runs_and_matches_the_package():
    from nerfart_amd import hip, scene
    ns = _doc_binding()
    model, _, _ = scene.build_model("VolSDF", seed=0, beta=0.01, device=DEV, precision="bf16x3")
    surf, rad = model.packed()
    assert ns["PREC"] == 1
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(1, 777, 3, generator=g) * 3 - 1.5).to(DEV)
    v = torch.nn.functional.normalize(torch.randn(1, 777, 3, generator=g), dim=-1).to(DEV)
    sdf_doc = ns["forward_surface"](surf, x, 3.0)
    rgb_doc, sdf2_doc, nab_doc = ns["forward"](surf, rad, x, v, 3.0)
    torch.cuda.synchronize()
    sdf = hip.sdf_fwd(surf, x.reshape(-1, 3).contiguous(), 3.0, precision=1)
    s2, nab, h7 = hip.sdf_nabla_fwd(surf, x.reshape(-1, 3).contiguous(), 3.0, precision=1)
    rgb = hip.radiance_fwd(rad, 1, x.reshape(-1, 3).contiguous(), v.reshape(-1, 3).contiguous(), nab, h7, precision=1)
    assert torch.equal(sdf_doc.reshape(-1), sdf) and torch.equal(sdf2_doc.reshape(-1), s2)
    assert torch.equal(nab_doc.reshape(-1, 3), nab) and torch.equal(rgb_doc.reshape(-1, 3), rgb)
    assert rgb_doc.shape == x.shape and sdf_doc.shape == x.shape[:-1]


def test_documented_clip_binding_runs_and_matches_the_package():
    """Section D's `EncodeImage` autograd function, executed as written, against clip_native.NativeImageEncoder (features and pixel
    gradient bit for bit: both are the same two C entry points)."""
    from nerfart_amd import hip, clip_vit, clip_native
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    code = next(b for b in blocks if "_nerfart_clip.py" in b and "class EncodeImage" in b)
    code = code.replace('C.CDLL("libnerfart_hip.so")', f'C.CDLL({hip.LIB_PATH!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md#D", "exec"), ns)
    model = clip_vit.build_clip(DEV, seed=0)
    blob = clip_native.pack_visual(model.state_dict(), DEV)
    g = torch.Generator().manual_seed(2)
    img = torch.randn(3, 3, 224, 224, generator=g).to(DEV)
    cot = torch.randn(3, 512, generator=g).to(DEV)
    a = img.clone().requires_grad_(True)
    fa = ns["EncodeImage"].apply(a, blob)
    fa.backward(cot)
    b = img.clone().requires_grad_(True)
    fb = clip_native.NativeImageEncoder(model)(b)
    fb.backward(cot)
    assert torch.equal(fa.detach(), fb.detach().float()) and torch.equal(a.grad, b.grad)


def test_documented_c_abi_only_render_binding_matches_the_trainer():
    """Section C's `HipRender` (volume_render forward + backward + the gradient fold on the C ABI alone), executed as written:
    its image equals the package's renderer, and its parameter gradients equal the package's thin wrapper over the same entry points
    bit for bit (ln_beta: the closed-form chain rule against autograd's)."""
    from nerfart_amd import hip, scene, rend_util, autodiff
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    code = next(b for b in blocks if "_nerfart_render.py" in b and "class HipRender" in b)
    code = code.replace('C.CDLL("libnerfart_hip.so")', f'C.CDLL({hip.LIB_PATH!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md#C", "exec"), ns)
    model, rk, render_fn = scene.build_model("VolSDF", seed=0, beta=0.01, device=DEV, precision="bf16x3")
    surf, rad = model.packed()
    H, W = 9, 7
    c2w, K = scene.camera(H, W)
    o, d, _ = rend_util.get_rays(c2w[None].to(DEV), K[None].to(DEV), H, W)
    o, d = o[0].contiguous(), d[0].contiguous()
    hr = ns["HipRender"](model, surf, rad, max_upsample_steps=rk["max_upsample_steps"], w_eikonal=0.1, patch_rays=20)
    rgb, depth, acc, d_all = hr.render(o, d)
    rgb_pkg, depth_pkg, ex = render_fn(o[None], d[None], calc_normal=False, detailed_output=True, require_nablas=True, **rk)
    # the binding lets the library build its own linspace tables (GPU formula); the package hands it torch's CPU tables: <= 1 ulp in depth
    assert float((rgb - rgb_pkg[0]).abs().max()) < 2e-5 and float((d_all - ex["d_vals"][0]).abs().max()) < 1e-5
    g = torch.rand(H * W, 3, generator=torch.Generator().manual_seed(4)).to(DEV) * 1e-2
    model.zero_grad()
    hr.backward(o[:40], d[:40], d_all[:40].contiguous(), g[:40].contiguous())
    hr.backward(o[40:], d[40:], d_all[40:].contiguous(), g[40:].contiguous())
    eik_doc = hr.step_grads()
    got = {n: p.grad.clone() for n, p in model.named_parameters()}
    assert float(hr.raw.abs().max()) == 0.0
    model.zero_grad()
    acc_ = autodiff.GradAccumulator()
    e1 = autodiff.volsdf_backward_samples_native(model, o[:40], d[:40], d_all[:40].contiguous(), g[:40], 0.1, True, False, accum=acc_, eik_group_rays=20)
    e2 = autodiff.volsdf_backward_samples_native(model, o[40:], d[40:], d_all[40:].contiguous(), g[40:], 0.1, True, False, accum=acc_, eik_group_rays=20)
    acc_.flush(model)
    assert abs(float(e1 + e2) - eik_doc) <= 1e-6 * abs(eik_doc)
    for n, p in model.named_parameters():
        if n == "ln_beta":
            assert torch.allclose(got[n], p.grad, rtol=1e-4, atol=1e-9), n
        else:
            assert torch.equal(got[n], p.grad), n
        assert float(p.grad.abs().max()) > 0, n


def test_documented_pack_binding_produces_the_package_blobs():
    """Section B's `_nerfart_pack.pack_blobs` (state-dict tensors -> blobs on the C ABI alone), executed as written, for the three precisions
    and both view embeddings: the blobs of the package's own models, bit for bit - and the section C binding can render and differentiate
    from them without importing anything of this repo."""
    from nerfart_amd import hip, scene
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    code = next(b for b in blocks if "_nerfart_pack.py" in b and "def pack_blobs" in b)
    code = code.replace('C.CDLL("libnerfart_hip.so")', f'C.CDLL({hip.LIB_PATH!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md#B-pack", "exec"), ns)
    for fw, vt in (("VolSDF", 1), ("NeuS", 3)):
        for precision, pid in (("fp32", 0), ("bf16x3", 1), ("fp16x2", 4)):
            model, _, _ = scene.build_model(fw, seed=0, beta=0.01 if fw == "VolSDF" else None, device=DEV, precision=precision)
            surf, rad = ns["pack_blobs"](model, precision=pid, view_tiles=vt)
            surf_pkg, rad_pkg = model.packed()
            torch.cuda.synchronize()
            assert torch.equal(surf.view(torch.int32), surf_pkg.view(torch.int32)), (fw, precision, "surface")
            assert torch.equal(rad.view(torch.int32), rad_pkg.view(torch.int32)), (fw, precision, "radiance")
