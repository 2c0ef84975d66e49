"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/nerfart_hip.h declares,
and its host helper reproduces torch.linspace bit for bit.  No compute calls (no GPU here)."""
import os
import re
import subprocess

import numpy as np
import torch

from conftest import REPO


def _header_symbols():
    src = open(os.path.join(REPO, "include", "nerfart_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nerfart_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from nerfart_amd import hip
    syms = _header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(hip.lib, s), f"{s} declared in include/nerfart_hip.h but not exported by {hip.LIB_PATH}"
    assert set(hip._SIGS) == set(syms), "ctypes signature table and header disagree"
    assert hip.ABI_VERSION == 5


def test_library_is_gfx950_code_object():
    from nerfart_amd import hip
    out = subprocess.run(["strings", "-n", "6", hip.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_linspace_helper():
    """nerfart_linspace implements the scalar ATen formula (the one torch's GPU kernel uses): step =
    (end-start)/(n-1), first half start + step*i, second half end - step*(n-1-i).  torch's CPU kernel is
    vectorised and may differ by one ulp on some entries, which is why the render entry points accept the
    host framework's own tables."""
    from nerfart_amd import hip
    for n in (2, 3, 16, 64, 66, 128, 512, 514, 2048):
        for a, b in ((0.0, 1.0), (0.0, 6.0), (-1.5, 2.25)):
            mine = hip.linspace(a, b, n).numpy()
            step = (np.float32(b) - np.float32(a)) / np.float32(n - 1)
            i = np.arange(n)
            ref = np.where(i < n // 2, np.float32(a) + step * i.astype(np.float32),
                           np.float32(b) - step * (n - 1 - i).astype(np.float32)).astype(np.float32)
            assert np.array_equal(mine, ref), (n, a, b)
            t = torch.linspace(a, b, n).numpy()
            assert np.max(np.abs(mine - t)) <= 2 * np.spacing(np.float32(max(abs(a), abs(b))))


def test_errors_are_reported_not_swallowed():
    from nerfart_amd import hip
    import pytest
    with pytest.raises(hip.NerfartHipError):
        hip.sdf_fwd(torch.zeros(8), torch.zeros(4, 3), 3.0)      # CPU tensors are refused: no CPU path


def test_new_entry_points_validate_their_arguments_without_a_gpu():
    """Argument checks of the round-2 entry points run before any launch: bad shapes / null buffers are errors with a message."""
    import ctypes as C
    from nerfart_amd import hip
    lib = hip.lib
    null = C.c_void_p(0)

    def err():
        return lib.nerfart_last_error().decode()
    assert lib.nerfart_gemm_f16_nt(null, null, 65, 64, 64, null, null) != 0 and "multiples of 64" in err()
    assert lib.nerfart_vgg16_l1_fwd(null, 0, null, 100, 100, null, 0, null, 0, null) != 0 and "H and W" in err()
    assert lib.nerfart_vgg16_l1_fwd(null, 0, null, 224, 224, null, 0, null, 0, null) != 0 and "workspace" in err()
    assert lib.nerfart_clip_vitb32_image_fwd(null, 0, null, 4, null, 1, null, 0, null) != 0 and "null" in err()
    assert lib.nerfart_clip_vitb32_workspace_bytes(0, 1) == 0
    assert lib.nerfart_clip_vitb32_workspace_bytes(16, 1) > 12 * 16 * 50 * 768 * 4
    assert lib.nerfart_resample_fwd(null, 2, 3, 8, 8, 0, 0, 8, 8, 4, 4, 1, null, null, null, null, 3, 4, 4, null) != 0 and "n_src" in err()
    assert lib.nerfart_resample_fwd(null, 1, 3, 8, 8, 0, 0, 8, 8, 4, 4, 7, null, null, null, null, 1, 4, 4, null) != 0 and "mode" in err()
    assert lib.nerfart_clip_style_heads(null, 17, null, null, null, null, 8, 79, 1.0, 0.2, 0.1, 2.0, 0.07, null, null, null) != 0 and "n_patches" in err()
    assert lib.nerfart_first_crossing(null, null, 5, 1, 0.0, null, null, null, null, null, null) != 0 and "n_steps" in err()
    assert lib.nerfart_first_crossing(null, null, 0, 256, 0.0, null, null, null, null, null, null) == 0          # no rays: nothing to do
    # round 3: the reverse-mode grad(SDF) scratch is the caller's (no hipMalloc inside the library)
    assert lib.nerfart_sdf_nabla_workspace_bytes(1) == 256 * 7 * 8 * 8 * 1024 and lib.nerfart_sdf_nabla_workspace_bytes(0) > 0
    assert lib.nerfart_sdf_nabla_workspace_bytes(2) == 0 and lib.nerfart_sdf_nabla_workspace_bytes(3) == 0
    one = C.c_void_p(16)                                           # a non-null, never dereferenced pointer: the checks come first
    assert lib.nerfart_sdf_nabla_fwd(one, 1, one, 4, 3.0, one, one, null, null, 0, null) != 0 and "workspace" in err()
    assert lib.nerfart_sdf_nabla_fwd(one, 1, one, 4, 3.0, one, one, null, one, 1024, null) != 0 and "workspace" in err()
    offs = (C.c_longlong * 22)()
    total = lib.nerfart_vgg16_blob_layout(C.cast(offs, C.c_void_p))
    assert offs[21] == total and total > 2 * 2 * (64 * 64 + 9 * (64 * 64 + 64 * 128 + 128 * 128 + 128 * 256 + 2 * 256 * 256))


def test_ctypes_signatures_have_the_headers_argument_counts_and_kinds():
    """Every prototype of include/nerfart_hip.h against hip._SIGS: same number of arguments, pointers where the header has pointers,
    64-bit integers where it says `long long`, floats where it says `float` - an argument added to the header (the workspace pair of
    nerfart_sdf_nabla_fwd, round 3) cannot be forgotten in the binding."""
    import ctypes as C
    from nerfart_amd import hip
    src = open(os.path.join(REPO, "include", "nerfart_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = re.findall(r"\b([a-z][a-z ]*?[\s\*]+)(nerfart_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S)
    assert len(protos) >= 55
    for ret, name, args in protos:
        args = " ".join(args.split())
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        restype, argtypes = hip._SIGS[name]
        assert len(argtypes) == len(params), f"{name}: header has {len(params)} arguments, hip._SIGS {len(argtypes)}"
        for p, t in zip(params, argtypes):
            if "*" in p:
                assert t is C.c_void_p or t is C.c_char_p, (name, p, t)
            elif p.startswith("long long") or p.startswith("unsigned long long") or p.startswith("size_t"):
                assert t is C.c_longlong, (name, p, t)
            elif p.startswith("float"):
                assert t is C.c_float, (name, p, t)
            elif p.startswith("int") or p.startswith("unsigned"):
                assert t is C.c_int, (name, p, t)


def test_render_bwd_entry_points_validate_their_arguments_without_a_gpu():
    from nerfart_amd import hip
    lib, null = hip.lib, None
    assert lib.nerfart_volsdf_render_bwd(null, null, 1, 6, null, null, 4, 192, null, null, null, null, null, null, null, 3.0, 100.0, 0.01, 0, 0.1, 0, 1,
                                         null, null, 0, null) != 0 and b"null" in lib.nerfart_last_error()
    assert lib.nerfart_volsdf_render_bwd(null, null, 1, 6, null, null, 20000, 192, null, null, null, null, null, null, null, 3.0, 100.0, 0.01, 0, 0.1, 0, 1,
                                         null, null, 0, null) != 0 and b"2^21" in lib.nerfart_last_error()
    assert lib.nerfart_neus_render_bwd(null, null, 2, 6, null, null, 4, 128, null, null, null, null, null, 20.0, 0, 0.1, 0, 0, null, null, 0, null) != 0
    assert b"view_tiles" in lib.nerfart_last_error()
    assert lib.nerfart_volsdf_render_bwd_workspace_bytes(1200, 192, 1) < lib.nerfart_volsdf_render_bwd_workspace_bytes(1200, 192, 0)


def test_an_architecture_the_kernels_are_not_written_for_is_refused_before_packing():
    """nerfart_pack_*_blob index their pointer tables and the device tensors with the fixed dims of the four shipped configs; a
    YAML-reachable variant (surface.W 512, D 6, skips [], W_geometry_feature 128, radiance W / D) must be a NotImplementedError at model
    construction - from the library's own dims (nerfart_pack_layer_dims) - not an out-of-bounds read on the GPU."""
    import pytest
    from nerfart_amd import hip, nets
    assert hip.pack_layer_dims(False, 6) == [(256, 39), (256, 256), (256, 256), (217, 256)] + [(256, 256)] * 4 + [(257, 256)]
    assert hip.pack_layer_dims(True, 1) == [(256, 265)] + [(256, 256)] * 3 + [(3, 256)]
    assert hip.pack_layer_dims(True, 3)[0] == (256, 289)
    with pytest.raises(NotImplementedError):
        hip.pack_layer_dims(False, 4)
    nets.VolSDF(W_geo_feat=256)                                                       # the shipped architecture constructs
    nets.NeuS(W_geo_feat=256, radiance_cfg={"embed_multires_view": 4})
    for bad in (dict(surface_cfg={"W": 512}), dict(surface_cfg={"D": 6}), dict(surface_cfg={"skips": []}), dict(surface_cfg={"skips": [3]}),
                dict(W_geo_feat=128), dict(radiance_cfg={"W": 128}), dict(radiance_cfg={"D": 3}), dict(surface_cfg={"embed_multires": 4})):
        with pytest.raises(NotImplementedError):
            nets.VolSDF(**dict(dict(W_geo_feat=256), **bad))
    # the pack wrappers hold the tensors themselves to the same dims (a state dict loaded into a hand-built module list)
    g = [torch.zeros(256, 1)] * 9; v = [torch.zeros(256, 256)] * 9; b = [torch.zeros(256)] * 9
    with pytest.raises(NotImplementedError):
        hip.pack_surface_blob(1, 6, g, v, b)
    with pytest.raises(NotImplementedError):
        hip.pack_surface_blob(1, 6, g[:7], v[:7], b[:7])


def test_rayschunk_is_read_as_the_memory_hint_it_is():
    """volsdf.launch_rays: the reference's rayschunk values (1024 / 2048 / 2000: render.py:488,614, volsdf.py:720,990) never shrink a launch below the
    library's own size unless the caller insists or the chunking is observable (perturb=True drawing from torch's generator per chunk)."""
    from nerfart_amd.volsdf import launch_rays, DEFAULT_RAYSCHUNK
    assert launch_rays(None, DEFAULT_RAYSCHUNK) == DEFAULT_RAYSCHUNK
    assert launch_rays(1024, DEFAULT_RAYSCHUNK) == DEFAULT_RAYSCHUNK and launch_rays(2048, DEFAULT_RAYSCHUNK) == DEFAULT_RAYSCHUNK
    assert launch_rays(1 << 20, DEFAULT_RAYSCHUNK) == 1 << 20                       # a LARGER request is honoured
    assert launch_rays(1024, DEFAULT_RAYSCHUNK, honor_rayschunk=True) == 1024
    assert launch_rays(1024, DEFAULT_RAYSCHUNK, perturb=True) == 1024                # draws per chunk: keep the caller's chunking
    assert launch_rays(1024, DEFAULT_RAYSCHUNK, perturb=True, uniforms=torch.zeros(1, 64)) == DEFAULT_RAYSCHUNK


def test_direct_more_refuses_a_bin_count_that_cannot_fit_the_lds():
    """N_nograd_samples is YAML-reachable (neus.py:735); ~10k+ bins per ray exceed the 160 KiB LDS of k_neus_upsample - refused up
    front with the limit in the message, not as a generic launch error from inside a render."""
    import ctypes as C
    from nerfart_amd import hip
    null = C.c_void_p(0)
    args = lambda nn: [null, null, 1, 3, null, null, 8, 1.0, 64.0, 64, 64, 4, 2, nn, 1 / 64., 0, 8192] + [null] * 3 + [0] + [null] * 12 + [null, 0, null]
    assert hip.lib.nerfart_neus_render_algo_fwd(*args(20000)) != 0 and "10,200" in hip.lib.nerfart_last_error().decode()
    assert hip.lib.nerfart_neus_render_algo_fwd(*args(2048)) != 0 and "10,200" not in hip.lib.nerfart_last_error().decode()      # passes THIS check (then: no workspace)
