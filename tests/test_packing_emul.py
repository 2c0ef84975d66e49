"""CPU validation of the weight-packing plan and of the HIP kernels' index arithmetic: a numpy model of
csrc/mlp_chain.hip (tests/emul_chain.py: MFMA lane layouts, chunk order, slot order, tangent quads)
walks the REAL packed blob and must reproduce the oracle."""
import numpy as np
import pytest
import torch

from conftest import scene_state
from oracle import nets
import emul_chain as em
from nerfart_amd import packing


def _blobs(fw):
    sd, _ = scene_state(fw, 0.01 if fw == "VolSDF" else None)
    surf = packing.surface_plan().pack(packing.surface_tensors(sd)).numpy()
    vt = 1 if fw == "VolSDF" else 3
    rad = packing.radiance_plan(vt).pack(packing.radiance_tensors(sd)).numpy()
    return sd, surf, rad, vt


def test_enc_slot_map_is_a_bijection_onto_39_features():
    feats = [packing.enc_slot_feature(s) for s in range(48)]
    real = [f for f in feats if f >= 0]
    assert sorted(real) == list(range(39)) and feats.count(-1) == 9


def test_blob_header_and_chunk_table():
    _, surf, rad, _ = _blobs("VolSDF")
    for blob, nc in ((surf, 59), (rad, 41)):
        hdr = blob[:512].view(np.int32)
        assert hdr[0] == packing.MAGIC and hdr[2] == nc and hdr[3] == blob.size
        nall = max(int(hdr[6]), nc)                        # the surface program carries its reverse-sweep chunks behind the forward ones
        assert (nall == 118) if blob is surf else (nall == nc)
        offs = hdr[16:16 + nall + 1]
        sizes = np.diff(offs)
        assert set(sizes[:nc].tolist()) <= {4096, 8192} and offs[0] == 512 and offs[-1] == hdr[4]
        assert set(sizes[nc:].tolist()) <= {8192, 6144}     # 6144: eight 3-tile k tiles of an encoding "tail"
    assert _blobs("NeuS")[2][:512].view(np.int32)[2] == 42


def test_emulated_sdf_only_matches_oracle():
    sd, surf, _, _ = _blobs("VolSDF")
    g = torch.Generator().manual_seed(7)
    pts = (torch.rand(16, 3, generator=g) * 6 - 3)
    pts[:4] *= 0.3
    ref = nets.volsdf_forward_surface(sd, pts)[0].numpy()
    out = em.emul_sdf_only(surf, pts.numpy(), 3.0)
    np.testing.assert_allclose(out, ref, atol=3e-6, rtol=1e-5)
    ref_nc = nets.surface_forward(sd, pts)[0].numpy()
    np.testing.assert_allclose(em.emul_sdf_only(surf, pts.numpy(), 0.0), ref_nc, atol=3e-6, rtol=1e-5)


def test_emulated_sdf_nabla_matches_oracle():
    sd, surf, _, _ = _blobs("VolSDF")
    g = torch.Generator().manual_seed(8)
    pts = (torch.rand(4, 3, generator=g) * 4 - 2)
    sdf, nab, h7 = em.emul_sdf_nabla(surf, pts.numpy(), 3.0)
    s_ref, n_ref, feat_ref = nets.surface_forward_with_nablas(sd, pts)
    d_bg = 3.0 - pts.norm(dim=-1)
    s_ref = torch.where(d_bg < s_ref, d_bg, s_ref)
    np.testing.assert_allclose(sdf, s_ref.numpy(), atol=3e-6, rtol=1e-5)
    np.testing.assert_allclose(nab, n_ref.numpy(), atol=2e-5, rtol=1e-4)
    # h7 -> feature via the last layer's rows 1..256 must give the oracle's geometry feature
    w8 = nets.folded_weight(sd, "implicit_surface.surface_fc_layers.8").numpy()
    b8 = sd["implicit_surface.surface_fc_layers.8.bias"].numpy()
    np.testing.assert_allclose(h7 @ w8[1:].T + b8[1:], feat_ref.numpy(), atol=1e-5, rtol=1e-4)


def test_emulated_reverse_mode_sdf_grad_matches_oracle():
    """k_sdf_grad's data flow (forward sweep keeping softplus', transposed chunks, 3-tile encoding tails, Jacobian of the encoding)
    on the packed blob, against autograd on the oracle's SDF net."""
    sd, surf, _, _ = _blobs("VolSDF")
    g = torch.Generator().manual_seed(12)
    pts = (torch.rand(16, 3, generator=g) * 4 - 2)
    pts[:3] *= 0.2
    sdf, nab, h7 = em.emul_sdf_grad(surf, pts.numpy(), 3.0)
    s_ref, n_ref, feat_ref = nets.surface_forward_with_nablas(sd, pts)
    d_bg = 3.0 - pts.norm(dim=-1)
    s_ref = torch.where(d_bg < s_ref, d_bg, s_ref)
    np.testing.assert_allclose(sdf, s_ref.numpy(), atol=3e-6, rtol=1e-5)
    np.testing.assert_allclose(nab, n_ref.numpy(), atol=2e-5, rtol=1e-4)
    w8 = nets.folded_weight(sd, "implicit_surface.surface_fc_layers.8").numpy()
    b8 = sd["implicit_surface.surface_fc_layers.8.bias"].numpy()
    np.testing.assert_allclose(h7 @ w8[1:].T + b8[1:], feat_ref.numpy(), atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("fw", ["VolSDF", "NeuS"])
def test_emulated_radiance_matches_oracle(fw):
    sd, surf, rad, vt = _blobs(fw)
    g = torch.Generator().manual_seed(9)
    pts = (torch.rand(16, 3, generator=g) * 2 - 1)
    view = torch.nn.functional.normalize(torch.randn(16, 3, generator=g), dim=-1)
    _, nab, feat = nets.surface_forward_with_nablas(sd, pts)
    # recover h7 from the oracle by running the hidden layers
    h7 = np.stack([em.emul_sdf_nabla(surf, pts[i:i + 4].numpy(), 0.0)[2] for i in range(0, 16, 4)]).reshape(16, 256)
    out = em.emul_radiance(rad, vt, pts.numpy(), view.numpy(), nab.numpy(), h7)
    ref = nets.radiance_forward(sd, pts, view, nab, feat, -1, -1 if fw == "VolSDF" else 4).numpy()
    np.testing.assert_allclose(out, ref, atol=3e-6, rtol=1e-5)


# ---- split-bf16 ("bf16x3") programs ---------------------------------------------------------------------
def _blobs_bf16(fw):
    sd, _ = scene_state(fw, 0.01 if fw == "VolSDF" else None)
    surf = packing.surface_plan_bf16().pack(packing.surface_tensors(sd)).numpy()
    vt = 1 if fw == "VolSDF" else 3
    rad = packing.radiance_plan_bf16(vt).pack(packing.radiance_tensors(sd)).numpy()
    return sd, surf, rad, vt


def test_bf16_blob_header():
    _, surf, rad, _ = _blobs_bf16("VolSDF")
    for blob, nc, nc_all, n_chunks in ((surf, 30, 59, 93), (rad, 21, 42, 42)):          # surface: 63 chunks + the 30 of the w32 program
        hdr = blob[:512].view(np.int32)
        assert hdr[0] == packing.MAGIC and hdr[2] == nc and hdr[6] == nc_all and hdr[3] == blob.size
        offs = hdr[16:16 + n_chunks + 1]
        # 1 or 2 k-steps of 32 KiB; the reverse programs hold one 48 KiB (8 k-steps x 3 tiles) / 16 KiB (x 1 tile) chunk
        assert set(np.diff(offs).tolist()) <= {4096, 8192, 12288, 16384} and offs[-1] == hdr[4]
        assert offs[-2] + 16384 <= blob.size          # the stream copies 64 KiB per chunk


def test_emulated_bf16_sdf_only_matches_oracle():
    sd, surf, _, _ = _blobs_bf16("VolSDF")
    g = torch.Generator().manual_seed(17)
    pts = (torch.rand(32, 3, generator=g) * 6 - 3)
    pts[:8] *= 0.3
    ref = nets.volsdf_forward_surface(sd, pts)[0].numpy()
    out = np.concatenate([em.emul_sdf_only_bf16(surf, pts[i:i + 16].numpy(), 3.0) for i in (0, 16)])
    np.testing.assert_allclose(out, ref, atol=1e-4, rtol=1e-4)


def test_emulated_bf16_sdf_nabla_matches_oracle():
    sd, surf, _, _ = _blobs_bf16("VolSDF")
    g = torch.Generator().manual_seed(18)
    pts = (torch.rand(8, 3, generator=g) * 4 - 2)
    parts = [em.emul_sdf_nabla_bf16(surf, pts[i:i + 4].numpy(), 3.0) for i in (0, 4)]
    sdf, nab, h7 = (np.concatenate([q[k] for q in parts]) for k in range(3))
    s_ref, n_ref, feat_ref = nets.surface_forward_with_nablas(sd, pts)
    d_bg = 3.0 - pts.norm(dim=-1)
    s_ref = torch.where(d_bg < s_ref, d_bg, s_ref)
    np.testing.assert_allclose(sdf, s_ref.numpy(), atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(nab, n_ref.numpy(), atol=5e-4, rtol=1e-3)
    w8 = nets.folded_weight(sd, "implicit_surface.surface_fc_layers.8").numpy()
    b8 = sd["implicit_surface.surface_fc_layers.8.bias"].numpy()
    np.testing.assert_allclose(h7 @ w8[1:].T + b8[1:], feat_ref.numpy(), atol=5e-4, rtol=1e-3)


@pytest.mark.parametrize("fw", ["VolSDF", "NeuS"])
def test_emulated_bf16_radiance_matches_oracle(fw):
    sd, surf, rad, vt = _blobs_bf16(fw)
    g = torch.Generator().manual_seed(19)
    pts = (torch.rand(32, 3, generator=g) * 2 - 1)
    view = torch.nn.functional.normalize(torch.randn(32, 3, generator=g), dim=-1)
    _, nab, feat = nets.surface_forward_with_nablas(sd, pts)
    h7 = np.concatenate([em.emul_sdf_nabla_bf16(surf, pts[i:i + 4].numpy(), 0.0)[2] for i in range(0, 32, 4)])
    out = np.concatenate([em.emul_radiance_bf16(rad, vt, pts[i:i + 16].numpy(), view[i:i + 16].numpy(), nab[i:i + 16].numpy(),
                                                h7[i:i + 16]) for i in (0, 16)])
    ref = nets.radiance_forward(sd, pts, view, nab, feat, -1, -1 if fw == "VolSDF" else 4).numpy()
    np.testing.assert_allclose(out, ref, atol=2e-4, rtol=1e-3)


def test_emulated_bf16_reverse_mode_grad_matches_oracle():
    """Forward program + transposed-weight backward program of the surface blob (k_sdf_grad_bf16's data flow)."""
    sd, surf, _, _ = _blobs_bf16("VolSDF")
    hdr = surf[:512].view(np.int32)
    assert hdr[2] == 30 and hdr[6] == 59
    g = torch.Generator().manual_seed(23)
    pts = (torch.rand(16, 3, generator=g) * 4 - 2)
    pts[:2] *= 2.0                                   # a few outside the bounding sphere (clamped sdf, nabla kept)
    sdf, nab, h7 = em.emul_sdf_grad_bf16(surf, pts.numpy(), 3.0)
    s_ref, n_ref, feat_ref = nets.surface_forward_with_nablas(sd, pts)
    d_bg = 3.0 - pts.norm(dim=-1)
    s_ref = torch.where(d_bg < s_ref, d_bg, s_ref)
    np.testing.assert_allclose(sdf, s_ref.numpy(), atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(nab, n_ref.numpy(), atol=2e-4, rtol=5e-4)
    w8 = nets.folded_weight(sd, "implicit_surface.surface_fc_layers.8").numpy()
    b8 = sd["implicit_surface.surface_fc_layers.8.bias"].numpy()
    np.testing.assert_allclose(h7 @ w8[1:].T + b8[1:], feat_ref.numpy(), atol=5e-4, rtol=1e-3)


@pytest.mark.parametrize("fw", ["VolSDF", "NeuS"])
def test_emulated_bf16_radiance_backward_matches_autograd(fw):
    """Transposed-weight program of the radiance blob (k_radiance_bwd_bf16's data flow) against torch autograd."""
    sd, surf, rad, vt = _blobs_bf16(fw)
    hdr = rad[:512].view(np.int32)
    assert hdr[2] == 21 and hdr[6] == 42
    g = torch.Generator().manual_seed(29)
    pts = (torch.rand(16, 3, generator=g) * 2 - 1)
    view = torch.nn.functional.normalize(torch.randn(16, 3, generator=g), dim=-1)
    _, nab, _ = nets.surface_forward_with_nablas(sd, pts)
    h7 = torch.rand(16, 256, generator=g) * 0.2
    W8 = nets.folded_weight(sd, "implicit_surface.surface_fc_layers.8")
    b8 = sd["implicit_surface.surface_fc_layers.8.bias"]
    mv = -1 if fw == "VolSDF" else 4
    h7g, nabg = h7.clone().requires_grad_(True), nab.clone().requires_grad_(True)
    feat = h7g @ W8[1:].T + b8[1:]
    # layer-by-layer forward to get the relu outputs
    from oracle.nets import embed, folded_weight
    x = torch.cat([pts, embed(view, mv), nabg, feat], dim=-1)
    acts = {}
    hcur = x
    for l in range(4):
        hcur = torch.relu(torch.nn.functional.linear(hcur, folded_weight(sd, f"radiance_net.layers.{l}"), sd[f"radiance_net.layers.{l}.bias"]))
        acts[l] = hcur
    rgb = torch.sigmoid(torch.nn.functional.linear(hcur, folded_weight(sd, "radiance_net.layers.4"), sd["radiance_net.layers.4.bias"]))
    np.testing.assert_allclose(rgb.detach().numpy(), nets.radiance_forward(sd, pts, view, nab, feat.detach(), -1, mv).numpy(), atol=1e-6)
    g_rgb = torch.randn(16, 3, generator=g)
    rgb.backward(g_rgb)
    g_h7, g_n, deltas = em.emul_radiance_bwd_bf16(rad, rgb.detach().numpy(), g_rgb.numpy(), {l: a.detach().numpy() for l, a in acts.items()})
    sc = float(h7g.grad.abs().max())
    np.testing.assert_allclose(g_h7, h7g.grad.numpy(), atol=2e-4 * sc, rtol=2e-3)
    np.testing.assert_allclose(g_n, nabg.grad.numpy(), atol=2e-4 * float(nabg.grad.abs().max()), rtol=2e-3)


# ---- the one-wave-per-SIMD K2 program (csrc/mlp_k2_w32.hip): v_mfma_f32_32x32x16_bf16 register path -------------------------
def _w32_frag_to_A(frag):
    """fragment [lane 64][e 8] -> A [32 rows][16 k]: lane l holds row l & 31, k = 8 (l >> 5) + e."""
    A = np.zeros((32, 16))
    for l in range(64):
        A[l & 31, 8 * (l >> 5): 8 * (l >> 5) + 8] = frag[l]
    return A


def _w32_unit_to_B(unit):
    """unit [lane 64][e 8] -> B [16 k][32 cols]: lane l holds column l & 31, k = 8 (l >> 5) + e."""
    B = np.zeros((16, 32))
    for l in range(64):
        B[8 * (l >> 5): 8 * (l >> 5) + 8, l & 31] = unit[l]
    return B


def _w32_C_to_regs(C):
    """C [32 rows][32 cols] -> [lane 64][reg 16]: lane (col = l & 31, h = l >> 5), reg r <-> row (r & 3) + 8 (r >> 2) + 4 h."""
    out = np.zeros((64, 16))
    for l in range(64):
        for r in range(16):
            out[l, r] = C[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    return out


def test_w32_program_register_path_matches_oracle():
    """Walks the fourth program of the split-bf16 surface blob exactly as mlp_k2_w32.hip does - fragments -> 32x32x16 MFMAs ->
    C registers -> (softplus) -> the next layer's B units straight from the registers - for one wave (32 columns)."""
    sd, _ = scene_state("VolSDF", 0.01)
    plan = packing.surface_plan_bf16()
    blob = plan.pack(packing.surface_tensors(sd))
    hdr = blob[:512].view(torch.int32).numpy()
    nc, first = int(hdr[8]), int(hdr[9])
    assert nc == 30 and first == 63
    offs = hdr[16 + first: 16 + first + nc + 1]
    aux = blob[hdr[4]: hdr[4] + hdr[5]].double().numpy()
    chunks = []
    for c in range(nc):
        body = blob[offs[c]: offs[c + 1]].view(torch.bfloat16).double().reshape(-1, 2, 64, 8)      # [item][term][lane][e]
        chunks.append((body[:, 0] + body[:, 1]).numpy())
    g = torch.Generator().manual_seed(9)
    pts = torch.rand(32, 3, generator=g) * 4 - 2
    pts[:6] *= 0.3
    x = pts.double().numpy()
    # encoding units as the kernel builds them: lane (col, h), slot (q, e) -> feature w32_feature_enc(q, h, e)
    e39 = nets.embed(pts, 6).double().numpy()
    enc = np.zeros((3, 64, 8))
    for q in range(3):
        for l in range(64):
            for e in range(8):
                f = packing.w32_feature_enc(q, l >> 5, e)
                enc[q, l, e] = e39[l & 31, f] if f >= 0 else 0.0
    softplus = lambda z: np.maximum(z, 0) + np.log1p(np.exp(-np.abs(100 * z))) / 100

    def bias_regs(l):
        b = aux[256 * l: 256 * l + 256]
        out = np.zeros((8, 64, 16))
        for T in range(8):
            for ln in range(64):
                for r in range(16):
                    out[T, ln, r] = b[32 * T + 8 * (r >> 2) + 4 * (ln >> 5) + (r & 3)]
        return out
    layer_ks = [3, 16, 16, 16, 17, 16, 16, 16]
    ci, P = 0, None
    for l in range(8):
        nks = layer_ks[l]
        nh = 0 if l == 0 else (14 if l == 4 else 16)
        Q = bias_regs(l)
        items = np.concatenate([chunks[ci + c] for c in range((nks + 3) // 4)])
        ci += (nks + 3) // 4
        assert items.shape[0] == nks * 8
        for ks in range(nks):
            if ks < nh:                                  # unit ks = softplus of regs 8 (ks & 1) .. of tile ks >> 1 of the previous layer
                unit = softplus(P[ks >> 1][:, 8 * (ks & 1): 8 * (ks & 1) + 8])
            else:
                unit = enc[ks - nh]
            B = _w32_unit_to_B(unit)
            for T in range(8):
                it = (ks % 4) * 8 + T + (ks // 4) * 32
                Q[T] += _w32_C_to_regs(_w32_frag_to_A(items[it]) @ B)
        P = Q
    row = aux[packing.SURF_AUX_ROW: packing.SURF_AUX_ROW + 256]
    sdf = np.zeros(32)
    for T in range(8):
        for ln in range(64):
            for r in range(16):
                sdf[ln & 31] += softplus(P[T][ln, r]) * row[32 * T + 8 * (r >> 2) + 4 * (ln >> 5) + (r & 3)]
    sdf += aux[packing.SURF_AUX_B8]
    ref = nets.surface_forward(sd, pts)[0].double().numpy()
    np.testing.assert_allclose(sdf, ref, atol=3e-5, rtol=1e-4)
    # the slot maps are bijections onto the features
    feats = sorted(packing.w32_feature_enc(q, h, e) for q in range(3) for h in range(2) for e in range(8))
    assert [f for f in feats if f >= 0] == list(range(39))
    assert sorted(packing.w32_feature_hidden(ks, h, e) for ks in range(16) for h in range(2) for e in range(8)) == list(range(256))


# ---- the 2-MFMA variant (C-ABI precision 4, csrc/mlp_chain_f16x2.hip): fp16 hi + lo weight fragments, one fp16 activation term ------------
def test_emulated_fp16x2_kernels_walk_the_fp16_blob():
    """The same programs packed with term="fp16" (fp16 fragments; the reverse chunks without the folded 1 / 65535) through the same
    emulated data flow with single-term fp16 activations: sdf / nabla / h7 / rgb land within the 11-bit-activation error of the
    oracle - a wrong fragment decode, a missing scale or an fp16 underflow of the transposed weights would miss by orders of magnitude."""
    sd, _ = scene_state("VolSDF", 0.01)
    surf = packing.surface_plan_bf16(term="fp16").pack(packing.surface_tensors(sd)).numpy()
    rad = packing.radiance_plan_bf16(1, term="fp16").pack(packing.radiance_tensors(sd)).numpy()
    g = torch.Generator().manual_seed(23)
    pts = (torch.rand(16, 3, generator=g) * 4 - 2)
    pts[:2] *= 2.0
    view = torch.nn.functional.normalize(torch.randn(16, 3, generator=g), dim=-1)
    em.TERM = "fp16"
    try:
        s_only = em.emul_sdf_only_bf16(surf, pts.numpy(), 3.0)
        sdf, nab, h7 = em.emul_sdf_grad_bf16(surf, pts.numpy(), 3.0)
        s_ref, n_ref, feat_ref = nets.surface_forward_with_nablas(sd, pts)
        rgb = em.emul_radiance_bf16(rad, 1, pts.numpy(), view.numpy(), n_ref.numpy(), h7)
    finally:
        em.TERM = "bf16"
    d_bg = 3.0 - pts.norm(dim=-1)
    s_ref = torch.where(d_bg < s_ref, d_bg, s_ref)
    err_s = np.abs(sdf - s_ref.numpy()).max()
    err_n = np.abs(nab - n_ref.numpy()).max()
    print(f"  fp16x2 emulation: sdf {err_s:.2e}, K2 sdf {np.abs(s_only - s_ref.numpy()).max():.2e}, nabla {err_n:.2e}")
    np.testing.assert_allclose(sdf, s_ref.numpy(), atol=3e-3)
    np.testing.assert_allclose(s_only, sdf, atol=1e-6)                        # K2 and the reverse-mode kernel's forward sweep: same arithmetic
    np.testing.assert_allclose(nab, n_ref.numpy(), atol=2e-2, rtol=2e-2)
    w8 = nets.folded_weight(sd, "implicit_surface.surface_fc_layers.8").numpy()
    b8 = sd["implicit_surface.surface_fc_layers.8.bias"].numpy()
    np.testing.assert_allclose(h7 @ w8[1:].T + b8[1:], feat_ref.numpy(), atol=2e-2, rtol=2e-2)
    ref = nets.radiance_forward(sd, pts, view, n_ref, feat_ref, -1, -1).numpy()
    np.testing.assert_allclose(rgb, ref, atol=1e-2)
    assert err_s > 1e-6, "suspiciously exact: is the emulation really on the fp16 path?"
