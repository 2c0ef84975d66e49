"""The weight-blob packers behind the C ABI (csrc/pack_blob.hip: nerfart_pack_surface_blob / nerfart_pack_radiance_blob - weight_norm fold,
unit-order permutation, hi / lo split on the device) against nerfart_amd/packing.py's numpy plans applied with torch (what the product used
until round 4 and what the CPU emulation walks): headers bit for bit; every weight the blob encodes - hi + lo of the split programs - to one
fp32 ulp as the encoding resolves it (measured: bit-identical blobs, the fold sums ||v|| in ATen's order), all programs, both networks' view
embeddings, the three precisions; then the renderer on a C-packed model vs the same model with torch-packed blobs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _decode(blob, plan, term):
    """(header int32[512], decoded chunk weights [n], aux floats) of a blob laid out by `plan`; split programs: hi + lo as fp32."""
    hdr = blob[:512].view(torch.int32).cpu().numpy()
    aux_off, n_aux = int(hdr[4]), int(hdr[5])
    body = blob[512:aux_off]
    if term == "fp32":
        w = body.clone()
    else:
        dt = torch.float16 if term == "fp16" else torch.bfloat16
        parts = body.reshape(-1, 2, 64, 4).view(dt).reshape(-1, 2, 64, 8).float()       # [unit][term][lane][e]
        w = (parts[:, 0] + parts[:, 1]).reshape(-1)
    return hdr, w, blob[aux_off:aux_off + n_aux], blob[aux_off + n_aux:]


@pytest.mark.parametrize("fw", ["VolSDF", "NeuS"])
@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "fp16x2"])
def test_c_packed_blobs_equal_the_numpy_plans(fw, precision):
    from nerfart_amd import scene, packing, hip
    model, _, _ = scene.build_model(fw, seed=0, beta=0.01 if fw == "VolSDF" else None, device=DEV, precision=precision)
    s, r = model.implicit_surface, model.radiance_net
    term = {"fp32": "fp32", "bf16x3": "bf16", "fp16x2": "fp16"}[precision]
    if precision == "fp32":
        sp, rp = packing.surface_plan(s.W, s.D, tuple(s.skips), s.embed_multires, s.W_geo_feat), packing.radiance_plan(model.view_tiles, r.W, r.D, s.W_geo_feat)
    else:
        sp = packing.surface_plan_bf16(s.W, s.D, tuple(s.skips), s.embed_multires, s.W_geo_feat, term=term)
        rp = packing.radiance_plan_bf16(model.view_tiles, r.W, r.D, s.W_geo_feat, term=term)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    with torch.no_grad():
        ref_blobs = (sp.pack(packing.surface_tensors(sd, D=s.D)), rp.pack(packing.radiance_tensors(sd, D_surf=s.D, D=r.D)))
    got_blobs = model.packed()
    for name, got, ref, plan in zip(("surface", "radiance"), got_blobs, ref_blobs, (sp, rp)):
        assert got.shape == ref.shape and got.nerfart_term == term, (name, got.shape, ref.shape)
        h_g, w_g, a_g, pad_g = _decode(got, plan, term)
        h_r, w_r, a_r, pad_r = _decode(ref, plan, term)
        np.testing.assert_array_equal(h_g, h_r, err_msg=f"{name} header")
        np.testing.assert_array_equal(h_g, plan.header, err_msg=f"{name} header vs plan")
        scale = float(w_r.abs().max())
        err = float((w_g - w_r).abs().max())
        identical = torch.equal(got.view(torch.int32), ref.view(torch.int32))
        print(f"  {fw} {precision} {name}: {w_r.numel()} chunk weights, max |diff| {err:.2e} (scale {scale:.2e}); blob bit-identical to the torch-packed one: {identical}")
        # the fold sums ||v|| in ATen's order (pack_blob.hip k_row_rnorm), so the blobs are expected bit-identical; the BOUND is what any fold of
        # the same weights must meet: one fp32 ulp of the weight, seen through the encoding - fp32 as is; a split blob's hi + lo resolves
        # 2^-16 (bf16) / 2^-21 (fp16, plus its subnormal spacing 6e-8 for |w| < 6e-5) of the weight
        res, floor = {"fp32": (3e-7, 0.0), "bf16": (2.0 ** -15, 1e-9), "fp16": (2.0 ** -20, 1.2e-7)}[term]
        bad = (w_g - w_r).abs() > res * w_r.abs() + floor
        assert not bool(bad.any()) and ((w_g == 0) == (w_r == 0)).all(), (name, int(bad.sum()), err)
        assert float(((a_g - a_r).abs() / (a_r.abs() + 1e-6)).max()) < 1e-6, name
        assert (pad_g == 0).all() and pad_g.numel() == pad_r.numel()


def test_sampler_blob_and_repack_after_an_update():
    """The mixed mode's second surface blob (fp16 hi + lo) comes from the same entry point; an in-place weight update re-packs both."""
    from nerfart_amd import scene
    model, rk, render_fn = scene.build_model("VolSDF", seed=0, beta=0.01, device=DEV, precision="mixed")
    b0, p0 = model.packed_sampler()
    assert p0 == 4 and b0.nerfart_term == "fp16" and model.packed()[0].nerfart_term == "bf16"
    assert model.packed_sampler()[0] is b0 and model.packed()[0] is model.packed()[0]
    before = [t.clone() for t in (b0,) + tuple(model.packed())]
    with torch.no_grad():
        model.implicit_surface.surface_fc_layers[2].weight_v.mul_(1.01)
    after = (model.packed_sampler()[0],) + tuple(model.packed())
    assert not torch.equal(before[0], after[0]) and not torch.equal(before[1], after[1])
    assert torch.equal(before[2][512:], after[2][512:]), "the radiance blob's content does not depend on the SDF net's hidden layers"


def test_bf16_only_entry_points_refuse_an_fp16_blob():
    from nerfart_amd import scene, hip
    model, _, _ = scene.build_model("VolSDF", seed=0, beta=0.01, device=DEV, precision="fp16x2")
    surf, rad = model.packed()
    o = torch.zeros(4, 3, device=DEV); d = torch.ones(4, 3, device=DEV); dep = torch.rand(4, 8, device=DEV).sort(-1)[0]
    with pytest.raises(hip.NerfartHipError, match="packed as fp16"):
        hip.volsdf_render_bwd(surf, rad, 1, 6, o, d, dep, torch.zeros(4, 3, device=DEV), hip.new_raw(DEV), R_bg=3.0, alpha=100.0, beta=0.01)
    # ... and so does the LIBRARY for a C-ABI caller (round 6): the packers record (blob pointer -> fragment encoding) on the host and the
    # entry points that read one encoding look the pointer up - no Python attribute involved (stripped here), no device read
    del surf.nerfart_term, rad.nerfart_term
    with pytest.raises(hip.NerfartHipError, match="packed as fp16"):
        hip.volsdf_render_bwd(surf, rad, 1, 6, o, d, dep, torch.zeros(4, 3, device=DEV), hip.new_raw(DEV), R_bg=3.0, alpha=100.0, beta=0.01)
    pts = torch.rand(64, 3, device=DEV)
    with pytest.raises(hip.NerfartHipError, match="split bf16"):
        hip.sdf_fwd(surf, pts, 3.0, precision=1)                       # an fp16 blob through the precision-1 kernels
    with pytest.raises(hip.NerfartHipError, match="fp32"):
        hip.sdf_fwd(surf, pts, 3.0, precision=0)
    hip.sdf_fwd(surf, pts, 3.0, precision=4)                           # its own precision: fine
    copy = surf.clone()                                                # a pointer the packers never saw is not checked (documented)
    hip.sdf_fwd(copy, pts, 3.0, precision=4)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "mixed"])
def test_render_on_c_packed_blobs_equals_render_on_torch_packed_blobs(precision):
    """End to end: 4,096 rays through the fused renderer with the C-packed blobs and with packing.py's - the weights agree to ~1 ulp, so do
    the pixels except where Algorithm 1 sits on a threshold (bounded like any other rounding change: tests/test_gpu_configs.pixel_budget)."""
    from nerfart_amd import scene, rend_util, packing
    from test_gpu_configs import pixel_budget
    model, rk, render_fn = scene.build_model("VolSDF", seed=0, beta=0.01, device=DEV, precision=precision)
    H = W = 64
    c2w, K = scene.camera(H, W)
    o, d, _ = rend_util.get_rays(c2w[None].to(DEV), K[None].to(DEV), H, W)
    kw = {k: v for k, v in rk.items() if k != "rayschunk"}
    rgb_c, _, ex_c = render_fn(o, d, require_nablas=True, calc_normal=True, detailed_output=True, **kw)
    s, r = model.implicit_surface, model.radiance_net
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    term = "bf16"
    if model.precision == "fp32":
        sp, rp = packing.surface_plan(), packing.radiance_plan(model.view_tiles)
    else:
        sp, rp = packing.surface_plan_bf16(term=term), packing.radiance_plan_bf16(model.view_tiles, term=term)
    with torch.no_grad():
        model._blobs = (sp.pack(packing.surface_tensors(sd, D=s.D)), rp.pack(packing.radiance_tensors(sd, D_surf=s.D, D=r.D)))
        if model.sampler_precision is not None:
            model._sampler_blob = (model._sampler_blob[0], packing.surface_plan_bf16(term="fp16").pack(packing.surface_tensors(sd, D=s.D)))
    rgb_t, _, ex_t = render_fn(o, d, require_nablas=True, calc_normal=True, detailed_output=True, **kw)
    same = ex_c["iter_usage"][0] == ex_t["iter_usage"][0]
    print(f"  {precision}: identical rounds on {float(same.float().mean()):.4f} of {H * W} rays")
    assert float(same.float().mean()) >= 0.99
    pixel_budget(rgb_c[0].cpu(), rgb_t[0].cpu(), f"C-packed vs torch-packed blobs ({precision})", stable=(same & (ex_t["iter_usage"][0] >= 0)).cpu(),
                 over_frac=1.5e-2, max_abs=2e-2, psnr_min=70.0)


def test_clip_blob_from_the_c_packer_equals_the_torch_statement():
    """nerfart_clip_vitb32_pack against the section-by-section torch statement of the blob (fp16 matrices stored once, fp32 vectors, zero
    padding), byte for byte; the library names the tensors it wants (nerfart_clip_vitb32_tensor_name)."""
    import ctypes as C
    from nerfart_amd import clip_vit, clip_native, hip
    model = clip_vit.build_clip(DEV, seed=0)
    state = model.state_dict()
    names = clip_native.tensor_names()
    assert len(names) == 152 and names[0] == ("conv1.weight", 768 * 3 * 32 * 32) and names[-1] == ("proj", 768 * 512)
    assert all(("visual." + n) in state and state["visual." + n].numel() == k for n, k in names)
    blob = clip_native.pack_visual(state, DEV)
    offs, total = clip_native.blob_layout()
    ref = torch.zeros(total, dtype=torch.uint8, device=DEV)

    def put(i, t, dtype):
        t = t.detach().to(DEV).to(dtype).contiguous().reshape(-1)
        ref[offs[i]: offs[i] + t.numel() * t.element_size()] = t.view(torch.uint8)
    g = lambda n: state["visual." + n]
    put(0, g("conv1.weight").reshape(768, -1), torch.float16)
    for l in range(12):
        p, s, f = f"transformer.resblocks.{l}.", 2 + 8 * l, 104 + 8 * l
        for j, n in ((0, "attn.in_proj_weight"), (2, "attn.out_proj.weight"), (4, "mlp.c_fc.weight"), (6, "mlp.c_proj.weight")):
            put(s + j, g(p + n), torch.float16)
        for j, n in enumerate(("ln_1.weight", "ln_1.bias", "attn.in_proj_bias", "attn.out_proj.bias", "ln_2.weight", "ln_2.bias", "mlp.c_fc.bias", "mlp.c_proj.bias")):
            put(f + j, g(p + n), torch.float32)
    put(99, g("proj"), torch.float16)
    for i, n in ((100, "class_embedding"), (101, "positional_embedding"), (102, "ln_pre.weight"), (103, "ln_pre.bias"), (200, "ln_post.weight"), (201, "ln_post.bias")):
        put(i, g(n), torch.float32)
    torch.cuda.synchronize()
    assert torch.equal(blob, ref)


def test_vgg_blob_from_the_c_packer_equals_the_torch_statement():
    import ctypes as C
    from nerfart_amd import vgg, hip
    m = vgg.VGGPerceptualLoss().to(DEV)
    blob = m.packed()
    offs = (C.c_longlong * 22)()
    total = hip.lib.nerfart_vgg16_blob_layout(C.cast(offs, C.c_void_p))
    ref = torch.zeros(total, dtype=torch.uint8, device=DEV)

    def put(i, t):
        t = t.detach().to(DEV).float().contiguous().reshape(-1)
        ref[offs[i]: offs[i] + 4 * t.numel()] = t.view(torch.uint8)
    for l, (idx, cin, cout, _) in enumerate(vgg._CONVS):
        w, b = m.net.features[str(idx)].weight.detach().float(), m.net.features[str(idx)].bias
        if l == 0:
            wf = torch.zeros(64, 64, device=w.device)
            wf[:, :27] = w.reshape(64, 27)
            put(0, wf[:, :32]); put(1, wf.t())
        else:
            put(3 * l, w.permute(0, 2, 3, 1).reshape(cout, 9 * cin))
            put(3 * l + 1, w.flip(2, 3).permute(1, 2, 3, 0).reshape(cin, 9 * cout))
        put(3 * l + 2, b)
    torch.cuda.synchronize()
    assert blob.numel() == total and torch.equal(blob, ref)
