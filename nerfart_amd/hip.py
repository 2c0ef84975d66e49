"""ctypes binding of libnerfart_hip.so (include/nerfart_hip.h) for PyTorch-ROCm tensors.

PyTorch supplies device memory and the current HIP stream; every arithmetic step of the hot path
runs in the library.  If the library is missing this module raises at import - there is no
fallback path.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NERFART_HIP_LIB", os.path.join(_HERE, "csrc", "libnerfart_hip.so"))

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the gfx950 kernels first "
        f"(python -c 'import __graft_entry__ as g; g.build()'  or  python -m nerfart_amd.build). "
        f"nerfart_amd has no CPU / eager-PyTorch fallback.")

lib = C.CDLL(LIB_PATH)


def csrc_sha256() -> str:
    """SHA-256 over the kernel sources (csrc/*.hip, *.h, *.cpp, sorted by name): stamps profiles so that a bench line can say
    whether a committed PMC summary was taken on the code it runs."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h"))
                    + glob.glob(os.path.join(_HERE, "csrc", "*.cpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()

_p, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong
_SIGS = {
    "nerfart_abi_version": (_i, []),
    "nerfart_last_error": (C.c_char_p, []),
    "nerfart_linspace": (None, [_f, _f, _i, _p]),
    "nerfart_profile_begin": (_i, []),
    "nerfart_profile_end": (_i, [_p, _p, _p]),
    "nerfart_profile_end5": (_i, [_p, _p, _p]),
    "nerfart_sdf_fwd": (_i, [_p, _i, _p, _ll, _f, _p, _p]),
    "nerfart_sdf_fwd_rays": (_i, [_p, _i, _p, _p, _p, _p, _i, _i, _i, _f, _p, _i, _p]),
    "nerfart_sdf_nabla_workspace_bytes": (_ll, [_i]),
    "nerfart_sdf_nabla_fwd": (_i, [_p, _i, _p, _ll, _f, _p, _p, _p, _p, _ll, _p]),
    "nerfart_sdf_nabla_fwd_rays": (_i, [_p, _i, _p, _p, _p, _p, _i, _i, _i, _f, _p, _p, _p, _p, _ll, _p]),
    "nerfart_wgrad_workspace_bytes": (_ll, [_i, _ll, _i]),
    "nerfart_wgrad_bf16": (_i, [_p, _ll, _p, _ll, _i, _ll, _i, _ll, _p, _p, _i, _p, _ll, _p]),
    "nerfart_ray_points": (_i, [_p, _p, _p, _ll, _i, _p, _p, _p]),
    "nerfart_volsdf_pass2_cotangents": (_i, [_p, _p, _p, _p, _p, _p, _ll, _i, _f, _f, _ll, _p, _p, _p, _p]),
    "nerfart_wgrad_operand_embed_pair": (_i, [_p, _p, _ll, _ll, _i, _p, _p]),
    "nerfart_wgrad_operand_inputs": (_i, [_p, _i, _p, _i, _p, _ll, _ll, _p, _p]),
    "nerfart_wgrad_operand_rgb_delta": (_i, [_p, _p, _ll, _ll, _p, _p, _p, _p]),
    "nerfart_wgrad_operand_sbar_ones": (_i, [_p, _ll, _ll, _p, _p]),
    "nerfart_radiance_fwd": (_i, [_p, _i, _i, _p, _p, _ll, _p, _p, _p, _p]),
    "nerfart_radiance_fwd_rays": (_i, [_p, _i, _i, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p]),
    "nerfart_get_rays": (_i, [_p, _p, _i, _i, _p, _i, _p, _p, _p]),
    "nerfart_normalize_dirs": (_i, [_p, _p, _i, _p]),
    "nerfart_linspace_depths": (_i, [_p, _i, _p, _p, _f, _f, _i, _p, _i, _p]),
    "nerfart_volsdf_first_check": (_i, [_i, _i, _i, _i, _f, _f, _f, _p, _p, _p, _i, _f, _p, _f, _p, _p, _p, _p, _p, _p, _p]),
    "nerfart_volsdf_upsample": (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p]),
    "nerfart_volsdf_merge_check": (_i, [_i, _i, _i, _i, _i, _i, _i, _f, _f, _f] + [_p] * 8 + [_i] + [_p] * 7),
    "nerfart_volsdf_finalize": (_i, [_i, _i, _i, _i] + [_p] * 4 + [_i] + [_p] * 5),
    "nerfart_volsdf_sampler_workspace_bytes": (_ll, [_i, _i, _i, _i, _i]),
    "nerfart_volsdf_fine_sample": (_i, [_p, _i, _p, _p, _i, _p, _p, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _ll, _p]),
    "nerfart_volsdf_fine_sample_guarded": (_i, [_p, _i, _p, _i, _f, _p, _p, _i, _p, _p, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _ll, _p]),
    "nerfart_volsdf_fine_sample_guarded2": (_i, [_p, _i, _p, _i, _f, _i, _p, _p, _i, _p, _p, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _ll, _p]),
    "nerfart_sort_concat": (_i, [_i, _p, _i, _i, _p, _i, _i, _p, _i, _p]),
    "nerfart_volsdf_composite": (_i, [_i, _i, _p, _p, _p, _p, _f, _f, _i] + [_p] * 8),
    "nerfart_volsdf_composite_bwd": (_i, [_i, _i, _p, _p, _p, _f, _f, _i, _p, _p, _p, _p, _p, _p]),
    "nerfart_neus_composite_bwd": (_i, [_i, _i, _p, _p, _f, _i, _p, _p, _p, _p, _p, _p]),
    "nerfart_sdf_fwd2_dump_bytes": (_ll, [_ll]),
    "nerfart_sdf_bwd2_dump_bytes": (_ll, [_ll]),
    "nerfart_sdf_fwd2": (_i, [_p, _p, _p, _ll, _p, _p]),
    "nerfart_sdf_bwd2": (_i, [_p, _ll, _p, _p, _p, _p, _p]),
    "nerfart_radiance_dump_bytes": (_ll, [_ll]),
    "nerfart_radiance_fwd_dump": (_i, [_p, _i, _p, _p, _ll, _p, _p, _p, _p, _p]),
    "nerfart_radiance_bwd": (_i, [_p, _ll, _p, _p, _p, _p, _p, _p, _p]),
    "nerfart_volsdf_render_workspace_bytes": (_ll, [_i, _i, _i, _i, _i]),
    "nerfart_volsdf_render_fwd": (_i, [_p, _p, _i, _i, _p, _p, _i, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i] + [_p] * 4 + [_i] + [_p] * 13 + [_p, _ll, _p]),
    "nerfart_volsdf_render_mixed_fwd": (_i, [_p, _p, _i, _p, _i, _i, _p, _p, _i, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i] + [_p] * 4 + [_i] + [_p] * 13 + [_p, _ll, _p]),
    "nerfart_volsdf_render_staged_fwd": (_i, [_p, _i, _p, _i, _p, _i, _f, _i, _p, _p, _i, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i] + [_p] * 4 + [_i] + [_p] * 13 + [_p, _p, _ll, _p]),
    "nerfart_volsdf_render_staged2_fwd": (_i, [_p, _i, _p, _i, _p, _i, _f, _i, _i, _p, _p, _i, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i] + [_p] * 4 + [_i] + [_p] * 13 + [_p, _p, _ll, _p]),
    "nerfart_near_far_from_sphere": (_i, [_p, _p, _i, _f, _p, _p, _p]),
    "nerfart_neus_upsample_step": (_i, [_i, _i, _i, _i, _f, _p, _p, _p, _i, _p, _p]),
    "nerfart_merge_sorted_pairs": (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "nerfart_neus_composite": (_i, [_i, _i, _p, _p, _p, _p, _f, _i] + [_p] * 9),
    "nerfart_neus_render_workspace_bytes": (_ll, [_i, _i, _i, _i]),
    "nerfart_neus_render_fwd": (_i, [_p, _p, _i, _i, _p, _p, _i, _f, _f, _i, _i, _i, _i, _i] + [_p] * 2 + [_i] + [_p] * 12 + [_p, _ll, _p]),
    "nerfart_neus_direct_upsample_step": (_i, [_i, _i, _i, _i, _f, _p, _p, _p, _i, _p, _p]),
    "nerfart_neus_render_algo_workspace_bytes": (_ll, [_i, _i, _i, _i, _i, _i]),
    "nerfart_neus_render_algo_fwd": (_i, [_p, _p, _i, _i, _p, _p, _i, _f, _f, _i, _i, _i, _i, _i, _f, _i, _i] + [_p] * 3 + [_i] + [_p] * 12 + [_p, _ll, _p]),
    "nerfart_clip_vitb32_blob_layout": (_ll, [_p]),
    "nerfart_clip_vitb32_workspace_bytes": (_ll, [_i, _i]),
    "nerfart_clip_vitb32_image_fwd": (_i, [_p, _ll, _p, _i, _p, _i, _p, _ll, _p]),
    "nerfart_clip_vitb32_image_bwd": (_i, [_p, _ll, _i, _p, _p, _p, _ll, _p]),
    "nerfart_gemm_f16_nt": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "nerfart_gemm_f16_nn": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "nerfart_vgg16_blob_layout": (_ll, [_p]),
    "nerfart_vgg16_workspace_bytes": (_ll, [_i, _i, _i]),
    "nerfart_vgg16_l1_fwd": (_i, [_p, _ll, _p, _i, _i, _p, _i, _p, _ll, _p]),
    "nerfart_vgg16_l1_bwd": (_i, [_p, _ll, _i, _i, _p, _p, _p, _ll, _p]),
    "nerfart_first_crossing": (_i, [_p, _p, _i, _i, _f, _p, _p, _p, _p, _p, _p]),
    "nerfart_secant_update": (_i, [_p, _i, _f, _p, _p, _p, _p]),
    "nerfart_root_finish": (_i, [_p, _p, _i, _p, _p, _p, _p, _f, _i, _p, _p, _p]),
    "nerfart_sphere_trace_step": (_i, [_p, _i, _p, _f, _p, _p, _p]),
    "nerfart_resample_fwd": (_i, [_p] + [_i] * 11 + [_p, _p, _p, _p, _i, _i, _i, _p]),
    "nerfart_resample_bwd": (_i, [_p] + [_i] * 11 + [_p, _p, _p, _p, _i, _i, _i, _p]),
    "nerfart_clip_style_heads": (_i, [_p, _i, _p, _p, _p, _p, _i, _i, _f, _f, _f, _f, _f, _p, _p, _p]),
    "nerfart_pass2_raw_layout": (_ll, [_p]),
    "nerfart_volsdf_render_bwd_workspace_bytes": (_ll, [_i, _i, _i]),
    "nerfart_volsdf_render_bwd": (_i, [_p, _p, _i, _i, _p, _p, _i, _i] + [_p] * 7 + [_f, _f, _f, _i, _f, _i, _i, _p, _p, _ll, _p]),
    "nerfart_neus_render_bwd_workspace_bytes": (_ll, [_i, _i, _i]),
    "nerfart_neus_render_bwd": (_i, [_p, _p, _i, _i, _p, _p, _i, _i] + [_p] * 5 + [_f, _i, _f, _i, _i, _p, _p, _ll, _p]),
    "nerfart_sdf_param_bwd_workspace_bytes": (_ll, [_ll]),
    "nerfart_sdf_param_bwd": (_i, [_p, _i, _p, _ll, _p, _p, _p, _p, _p, _ll, _p]),
    "nerfart_radiance_param_bwd_workspace_bytes": (_ll, [_ll]),
    "nerfart_radiance_param_bwd": (_i, [_p, _i, _p, _p, _p, _p, _ll, _p, _p, _p, _p, _i, _p, _p, _ll, _p]),
    "nerfart_folded_grads_layout": (_ll, [_i, _i, _p]),
    "nerfart_fold_weight_grads": (_i, [_p, _i, _i, _p, _p]),
    "nerfart_weight_norm_bwd": (_i, [_p, _p, _p, _i, _i, _p, _p, _i, _p]),
    "nerfart_surface_blob_floats": (_ll, [_i, _i]),
    "nerfart_radiance_blob_floats": (_ll, [_i, _i]),
    "nerfart_pack_workspace_bytes": (_ll, []),
    "nerfart_pack_layer_dims": (_i, [_i, _i, _p, _p]),
    "nerfart_geometry_feature_workspace_bytes": (_ll, []),
    "nerfart_geometry_feature": (_i, [_p, _p, _p, _p, _ll, _p, _p, _ll, _p]),
    "nerfart_pack_surface_blob": (_i, [_i, _i, _p, _p, _p, _p, _ll, _p, _ll, _p]),
    "nerfart_pack_radiance_blob": (_i, [_i, _i, _p, _p, _p, _p, _p, _p, _p, _ll, _p, _ll, _p]),
    "nerfart_pack_plan_debug": (_i, [_i, _i, _i] + [_p] * 6),
    "nerfart_clip_vitb32_n_tensors": (_i, []),
    "nerfart_clip_vitb32_tensor_name": (_ll, [_i, _p, _i]),
    "nerfart_clip_vitb32_pack": (_i, [_p, _p, _ll, _p]),
    "nerfart_vgg16_pack": (_i, [_p, _p, _p, _ll, _p]),
}
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)          # AttributeError here = header / library mismatch
    _fn.restype, _fn.argtypes = _res, _args

ABI_VERSION = lib.nerfart_abi_version()


class NerfartHipError(RuntimeError):
    pass


def _check(rc: int, what: str):
    if rc != 0:
        raise NerfartHipError(f"{what} failed (code {rc}): {lib.nerfart_last_error().decode(errors='replace')}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, dtype=torch.float32, name="tensor") -> int:
    if t is None:
        return None
    if not t.is_cuda:
        raise NerfartHipError(f"{name} must live on the GPU (got {t.device}); nerfart_amd has no CPU path")
    if t.dtype != dtype:
        raise NerfartHipError(f"{name} must be {dtype} (got {t.dtype})")
    if not t.is_contiguous():
        raise NerfartHipError(f"{name} must be contiguous")
    return t.data_ptr()


def _ptr_table(tensors, name):
    """HOST array of device pointers (what the pack entry points take for per-layer tensors)."""
    return (C.c_void_p * len(tensors))(*[_dev(t, name=f"{name}[{i}]") for i, t in enumerate(tensors)])


def pack_layer_dims(radiance: bool, arg: int):
    """[(rows, cols)] of the weight_v tensors the packers expect (nerfart_pack_layer_dims): the SDF net for embed_multires = arg, or the
    radiance net for view_tiles = arg."""
    rows, cols = (C.c_int * 9)(), (C.c_int * 9)()
    n = int(lib.nerfart_pack_layer_dims(int(bool(radiance)), int(arg), rows, cols))
    if n <= 0:
        raise NotImplementedError(f"nerfart_pack_layer_dims: {lib.nerfart_last_error().decode(errors='replace')}")
    return [(int(rows[l]), int(cols[l])) for l in range(n)]


def check_layers(what: str, dims, weight_g, weight_v, bias):
    """The pack entry points index HOST pointer tables and DEVICE tensors with the fixed dims of the architecture the kernels are written for:
    refuse anything else here (NotImplementedError, as the reference-config guard of the packing plans did) instead of reading out of bounds."""
    if not (len(weight_g) == len(weight_v) == len(bias) == len(dims)):
        raise NotImplementedError(f"{what}: the kernels are written for {len(dims)} layers (got {len(weight_g)} / {len(weight_v)} / {len(bias)} "
                                  f"weight_g / weight_v / bias tensors) - SURVEY.md 2: the four shipped configs")
    for l, (r, c) in enumerate(dims):
        if tuple(weight_v[l].shape) != (r, c) or weight_g[l].numel() != r or tuple(bias[l].shape) != (r,):
            raise NotImplementedError(f"{what}: layer {l} must be weight_v [{r}, {c}], weight_g [{r}, 1], bias [{r}] (got {tuple(weight_v[l].shape)}, "
                                      f"{tuple(weight_g[l].shape)}, {tuple(bias[l].shape)}): W 256, D 8, skips [4], embed_multires 6, W_geo_feat 256 / "
                                      f"radiance W 256, D 4 are the architectures the kernels are written for")


def pack_surface_blob(precision: int, multires: int, weight_g, weight_v, bias) -> torch.Tensor:
    """nerfart_pack_surface_blob: the SDF net's blob for C-ABI `precision` (0 fp32, 1 split bf16, 4 fp16 hi + lo) from the state dict's
    per-layer weight_g [out, 1] / weight_v [out, in] / bias [out] (9 layers) - weight_norm fold, unit-order permutation and hi / lo split on
    the device.  The returned tensor carries `.nerfart_term` ('fp32' | 'bf16' | 'fp16') for the wrappers of entry points that read one encoding only."""
    check_layers("pack_surface_blob", pack_layer_dims(False, multires), weight_g, weight_v, bias)
    dev = weight_v[0].device
    n = int(lib.nerfart_surface_blob_floats(int(precision), int(multires)))
    if n <= 0:
        raise NerfartHipError(f"nerfart_surface_blob_floats: {lib.nerfart_last_error().decode(errors='replace')}")
    blob = torch.empty(n, dtype=torch.float32, device=dev)
    ws = _workspace(int(lib.nerfart_pack_workspace_bytes()), dev)
    g = [t.detach().reshape(-1).contiguous() for t in weight_g]
    v = [t.detach().contiguous() for t in weight_v]
    b = [t.detach().contiguous() for t in bias]
    _check(lib.nerfart_pack_surface_blob(int(precision), int(multires), _ptr_table(g, "weight_g"), _ptr_table(v, "weight_v"), _ptr_table(b, "bias"),
                                         _dev(blob), n, ws.data_ptr(), ws.numel(), _stream()), "nerfart_pack_surface_blob")
    blob.nerfart_term = {0: "fp32", 1: "bf16", 4: "fp16", 5: "fp16 (softplus-scaled, precision 5)"}[int(precision)]
    return blob


def pack_radiance_blob(precision: int, view_tiles: int, surf8, weight_g, weight_v, bias) -> torch.Tensor:
    """nerfart_pack_radiance_blob: surf8 = (weight_g, weight_v, bias) of the SDF net's last layer (its rows 1.. make the geometry feature),
    then the 5 radiance layers' tensors."""
    check_layers("pack_radiance_blob", pack_layer_dims(True, view_tiles), weight_g, weight_v, bias)
    check_layers("pack_radiance_blob (the SDF net's last layer)", pack_layer_dims(False, 6)[8:], [surf8[0]], [surf8[1]], [surf8[2]])
    dev = weight_v[0].device
    n = int(lib.nerfart_radiance_blob_floats(int(precision), int(view_tiles)))
    if n <= 0:
        raise NerfartHipError(f"nerfart_radiance_blob_floats: {lib.nerfart_last_error().decode(errors='replace')}")
    blob = torch.empty(n, dtype=torch.float32, device=dev)
    ws = _workspace(int(lib.nerfart_pack_workspace_bytes()), dev)
    s8 = [surf8[0].detach().reshape(-1).contiguous(), surf8[1].detach().contiguous(), surf8[2].detach().contiguous()]
    g = [t.detach().reshape(-1).contiguous() for t in weight_g]
    v = [t.detach().contiguous() for t in weight_v]
    b = [t.detach().contiguous() for t in bias]
    _check(lib.nerfart_pack_radiance_blob(int(precision), int(view_tiles), _dev(s8[0], name="surf8_g"), _dev(s8[1], name="surf8_v"), _dev(s8[2], name="surf8_bias"),
                                          _ptr_table(g, "weight_g"), _ptr_table(v, "weight_v"), _ptr_table(b, "bias"), _dev(blob), n, ws.data_ptr(), ws.numel(),
                                          _stream()), "nerfart_pack_radiance_blob")
    blob.nerfart_term = {0: "fp32", 1: "bf16", 4: "fp16"}[int(precision)]
    return blob


def need_term(blob, term: str, who: str):
    """Entry points that read ONE fragment encoding (the training kernels: split bf16) refuse a blob packed for another - the sizes and the
    program id of a bf16 and an fp16 blob coincide, only header word 10 differs."""
    have = getattr(blob, "nerfart_term", None)
    if have is not None and have != term:
        raise NerfartHipError(f"{who} reads {term} blobs; this blob was packed as {have}")


def linspace(start: float, end: float, n: int) -> torch.Tensor:
    """Host helper: the library's restatement of torch.linspace (used for its own tables)."""
    out = torch.empty(n, dtype=torch.float32)
    lib.nerfart_linspace(start, end, n, out.data_ptr())
    return out


def profile_begin():
    lib.nerfart_profile_begin()


def profile_end():
    """-> {kernel: (ms, launches, units)} for the chained-MLP kernels (units = points) and k_wgrad<256> (units = algorithmic bytes)
    launched since profile_begin(); k_sdf_only_escalation: the SDF queries of the guarded sampler's second run (the re-sampled rays, at the
    escalation precision) - kept apart from k_sdf_only, the dominant kernel's own launches."""
    ms = (C.c_double * 5)(); ln = (C.c_longlong * 5)(); un = (C.c_longlong * 5)()
    lib.nerfart_profile_end5(ms, ln, un)
    return {k: (ms[i], ln[i], un[i]) for i, k in enumerate(("k_sdf_only", "k_sdf_nabla", "k_radiance", "k_wgrad256", "k_sdf_only_escalation"))}


# ---- point queries -----------------------------------------------------------------------
PRECISIONS = {"fp32": 0, "bf16x3": 1, "fp16x2": 4}      # 4: the 2-MFMA kernels: the sampler of the shipped "mixed" mode (nets.set_precision); everywhere: a measurement variant
# Precisions that exist for Algorithm 1's no-gradient SDF queries ONLY (nerfart_sdf_fwd[_rays] and the sampler stage of the renderers refuse them
# nowhere else): 5 = "fp16x1", one MFMA per product on the precision-4 blob (csrc/mlp_chain_f16x1.hip).
SAMPLER_PRECISIONS = dict(PRECISIONS, fp16x1=5, fp16x1c=5)     # fp16x1c: the same kernel on error-compensated one-term weights (nerfart_amd/calibrate.py)


def sdf_fwd(surf_blob, pts, R_bg: float, precision: int = 0):
    M = pts.shape[0]
    out = torch.empty(M, dtype=torch.float32, device=pts.device)
    _check(lib.nerfart_sdf_fwd(_dev(surf_blob, name="surf_blob"), int(precision), _dev(pts, name="pts"), M, float(R_bg), _dev(out), _stream()),
           "nerfart_sdf_fwd")
    return out


def sdf_nabla_fwd(surf_blob, pts, R_bg: float, want_h7: bool = True, precision: int = 0):
    M = pts.shape[0]
    sdf = torch.empty(M, dtype=torch.float32, device=pts.device)
    nab = torch.empty(M, 3, dtype=torch.float32, device=pts.device)
    h7 = torch.empty(M, 256, dtype=torch.float32, device=pts.device) if want_h7 else None
    ws, nb = nabla_workspace(precision, pts.device)
    _check(lib.nerfart_sdf_nabla_fwd(_dev(surf_blob), int(precision), _dev(pts, name="pts"), M, float(R_bg), _dev(sdf), _dev(nab),
                                     _dev(h7), _dev(ws, torch.uint8), nb, _stream()), "nerfart_sdf_nabla_fwd")
    return sdf, nab, h7


def geometry_feature(weight_g, weight_v, bias, h7):
    """feat [M, 256] = W8[1:] h7 + b8[1:] (nerfart_geometry_feature: the SDF net's last layer's feature rows, weight_norm folded on the device)."""
    M = h7.shape[0]
    out = torch.empty(M, 256, dtype=torch.float32, device=h7.device)
    ws = _workspace(int(lib.nerfart_geometry_feature_workspace_bytes()), h7.device)
    _check(lib.nerfart_geometry_feature(_dev(weight_g.detach().reshape(-1).contiguous(), name="weight_g"), _dev(weight_v.detach().contiguous(), name="weight_v"),
                                        _dev(bias.detach().contiguous(), name="bias"), _dev(h7, name="h7"), M, _dev(out), ws.data_ptr(), ws.numel(), _stream()),
           "nerfart_geometry_feature")
    return out


def nabla_workspace(precision: int, device):
    """The reverse-mode grad(SDF) kernels' scratch (softplus' of the forward sweep), from PyTorch's caching allocator: visible
    to torch.cuda.max_memory_allocated(), stream-ordered like every other tensor of the call."""
    nb = int(lib.nerfart_sdf_nabla_workspace_bytes(int(precision)))
    return (torch.empty(nb, dtype=torch.uint8, device=device) if nb else None), nb


def wgrad(Z, A, n_mats: int, rows: int, a_cols: int, z_stride: int, a_stride: int, cs_rows: int = 0, want_cs: bool = False):
    """dW [n_mats, 256, a_cols] fp32 = Z_m^T A_m (+ cs [n_mats, 256] = column sums of the first cs_rows rows of Z_m) over bf16
    row-major operands: Z / A are tensors whose data_ptr() is matrix 0 ([rows, 256] / [rows, a_cols]); strides in BYTES between
    matrices (0 = shared operand).  nerfart_wgrad_bf16 (csrc/wgrad.hip)."""
    dev = Z.device
    assert Z.dtype == torch.bfloat16 and A.dtype == torch.bfloat16 and Z.is_cuda and A.is_cuda
    dW = torch.empty(n_mats, 256, a_cols, dtype=torch.float32, device=dev)
    cs = torch.empty(n_mats, 256, dtype=torch.float32, device=dev) if want_cs else None
    nb = int(lib.nerfart_wgrad_workspace_bytes(n_mats, rows, a_cols))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    _check(lib.nerfart_wgrad_bf16(Z.data_ptr(), int(z_stride), A.data_ptr(), int(a_stride), n_mats, rows, a_cols, int(cs_rows),
                                  _dev(dW), _dev(cs), 0, _dev(ws, torch.uint8), nb, _stream()), "nerfart_wgrad_bf16")
    return dW, cs


# ---- per-point glue of pass 2 (csrc/pass2_operands.hip) ------------------------------------------
def ray_points(rays_o, rays_dn, depth, want_view: bool = True):
    """pts [R P, 3] = o + dn * depth, view [R P, 3] = dn per sample."""
    R, P = depth.shape
    pts = torch.empty(R * P, 3, dtype=torch.float32, device=depth.device)
    view = torch.empty(R * P, 3, dtype=torch.float32, device=depth.device) if want_view else None
    _check(lib.nerfart_ray_points(_dev(rays_o, name="rays_o"), _dev(rays_dn, name="rays_dn"), _dev(depth, name="depth"), R, P, _dev(pts),
                                  _dev(view), _stream()), "nerfart_ray_points")
    return pts, view


def volsdf_pass2_cotangents(pts, sdf, g_sdf, nabla, g_n, R: int, P: int, R_bg: float, w_eikonal: float, eik_group_rays=None, g_n_extra=None):
    """(sbar [R P], nbar [R P, 3], eik_ray [R]) - see include/nerfart_hip.h."""
    dev = pts.device
    sbar = torch.empty(R * P, dtype=torch.float32, device=dev)
    nbar = torch.empty(R * P, 3, dtype=torch.float32, device=dev)
    eik_ray = torch.empty(R, dtype=torch.float32, device=dev)
    _check(lib.nerfart_volsdf_pass2_cotangents(_dev(pts, name="pts"), _dev(sdf, name="sdf"), _dev(g_sdf, name="g_sdf"), _dev(nabla, name="nabla"),
                                               _dev(g_n, name="g_n"), _dev(g_n_extra, name="g_n_extra"), R, P, float(R_bg), float(w_eikonal),
                                               int(eik_group_rays or 0), _dev(sbar), _dev(nbar), _dev(eik_ray), _stream()),
           "nerfart_volsdf_pass2_cotangents")
    return sbar, nbar, eik_ray


def _operand(rows: int, device):
    return torch.empty(rows, 64, dtype=torch.bfloat16, device=device)


def wgrad_operand_embed_pair(pts, direction, rows_pad: int, multires: int):
    out = _operand(2 * rows_pad, pts.device)
    _check(lib.nerfart_wgrad_operand_embed_pair(_dev(pts, name="pts"), _dev(direction, name="dir"), pts.shape[0], rows_pad, int(multires),
                                                out.data_ptr(), _stream()), "nerfart_wgrad_operand_embed_pair")
    return out


def wgrad_operand_inputs(x, multires_x: int, view, multires_view: int, normals, rows_pad: int):
    out = _operand(rows_pad, x.device)
    _check(lib.nerfart_wgrad_operand_inputs(_dev(x, name="x"), int(multires_x), _dev(view, name="view"), int(multires_view),
                                            _dev(normals, name="normals"), x.shape[0], rows_pad, out.data_ptr(), _stream()),
           "nerfart_wgrad_operand_inputs")
    return out


def wgrad_operand_rgb_delta(rgb, g_rgb, rows_pad: int, want_d4: bool = False):
    """(operand [rows_pad, 64] bf16, column sums of d4 = g_rgb rgb (1 - rgb) [3], d4 [M, 3] or None)"""
    out = _operand(rows_pad, rgb.device)
    d4 = torch.empty(rgb.shape[0], 3, dtype=torch.float32, device=rgb.device) if want_d4 else None
    sums = torch.empty((rows_pad + 31) // 32, 3, dtype=torch.float32, device=rgb.device)
    _check(lib.nerfart_wgrad_operand_rgb_delta(_dev(rgb, name="rgb"), _dev(g_rgb, name="g_rgb"), rgb.shape[0], rows_pad, out.data_ptr(),
                                               _dev(d4), _dev(sums), _stream()), "nerfart_wgrad_operand_rgb_delta")
    return out, sums.sum(0), d4


def wgrad_operand_sbar_ones(sbar, rows_pad: int):
    out = _operand(2 * rows_pad, sbar.device)
    _check(lib.nerfart_wgrad_operand_sbar_ones(_dev(sbar, name="sbar"), sbar.shape[0], rows_pad, out.data_ptr(), _stream()),
           "nerfart_wgrad_operand_sbar_ones")
    return out


def radiance_fwd(rad_blob, view_tiles: int, pts, view, nabla, h7, precision: int = 0):
    M = pts.shape[0]
    rgb = torch.empty(M, 3, dtype=torch.float32, device=pts.device)
    _check(lib.nerfart_radiance_fwd(_dev(rad_blob), int(precision), int(view_tiles), _dev(pts, name="pts"), _dev(view, name="view"), M,
                                    _dev(nabla, name="nabla"), _dev(h7, name="h7"), _dev(rgb), _stream()),
           "nerfart_radiance_fwd")
    return rgb


def sdf_fwd_rays(surf_blob, rays_o, rays_dn, depth, R_bg: float, ray_idx=None, n_per_ray=None, precision: int = 0):
    """depth [n_slots, stride >= n_per_ray] -> sdf [n_slots, n_per_ray]"""
    n_slots, stride = depth.shape
    n_per_ray = stride if n_per_ray is None else n_per_ray
    out = torch.empty(n_slots, n_per_ray, dtype=torch.float32, device=depth.device)
    _check(lib.nerfart_sdf_fwd_rays(_dev(surf_blob), int(precision), _dev(rays_o), _dev(rays_dn), _dev(ray_idx, torch.int32),
                                    _dev(depth), n_slots, n_per_ray, stride, float(R_bg), _dev(out), n_per_ray, _stream()),
           "nerfart_sdf_fwd_rays")
    return out


# ---- rays ------------------------------------------------------------------------------------
def get_rays(pose, K, H: int, W: int, select=None):
    n = H * W if select is None else select.numel()
    o = torch.empty(n, 3, dtype=torch.float32, device=pose.device)
    d = torch.empty(n, 3, dtype=torch.float32, device=pose.device)
    _check(lib.nerfart_get_rays(_dev(pose, name="pose"), _dev(K, name="K"), H, W, _dev(select, torch.int64), n, _dev(o), _dev(d),
                                _stream()), "nerfart_get_rays")
    return o, d


def normalize_dirs(d):
    out = torch.empty_like(d)
    _check(lib.nerfart_normalize_dirs(_dev(d), _dev(out), d.shape[0], _stream()), "nerfart_normalize_dirs")
    return out


_lin_cache = {}


def lin_table(n: int, device) -> torch.Tensor:
    """torch.linspace(0, 1, n) as the reference builds it (CPU kernel, then moved; volsdf.py:472,483), cached."""
    key = (n, str(device))
    if key not in _lin_cache:
        _lin_cache[key] = torch.linspace(0, 1, n).float().to(device)
    return _lin_cache[key]


_ws_cache = {}
_ws_lock = __import__("threading").Lock()


def _workspace(nbytes: int, device) -> torch.Tensor:
    """One cached byte buffer per (device, stream), grown on demand (288 GB of HBM: keep it resident).  Per stream: the entry points
    run asynchronously on torch's current stream, so two streams (or two host threads on their own streams) must not share scratch."""
    key = (str(device), int(torch.cuda.current_stream(device).cuda_stream))
    with _ws_lock:                                                 # two host threads on their own streams may grow / evict at once
        ws = _ws_cache.get(key)
        if ws is None or ws.numel() < nbytes:
            _ws_cache.pop(key, None)
            while len(_ws_cache) >= 8:                             # short-lived streams must not pin memory for good
                _ws_cache.pop(next(iter(_ws_cache)), None)
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            _ws_cache[key] = ws
    return ws


def volsdf_fine_sample(surf_blob, rays_o, rays_dn, near: float, far: float, R_bg: float, alpha: float, beta: float,
                       eps: float, n_init: int, n_up: int, n_final: int, max_iter: int, max_bisect: int, precision: int = 0,
                       u_final=None, escalate=None, guard: float = 0.0, stats=None, late_round: int = 0):
    """u_final [R, n_final]: the caller's uniform random numbers for the final inverse-CDF samples (perturb=True:
    sample_cdf(det=False), rend_util.py:306-307); None: the deterministic linspace table.
    escalate = (surface blob, precision id) + guard > 0: the GUARDED sampler (nerfart_volsdf_fine_sample_guarded): rays whose convergence decision
    lies within guard * eps of eps, and rays that never converge, are sampled again on that blob.  stats (dict): 'escalated' += rays that ran twice.
    late_round > 0 (with the guard): rays still active after that up-sampling round are escalated there (nerfart_volsdf_fine_sample_guarded2)."""
    R = rays_o.shape[0]
    dev = rays_o.device
    if u_final is not None and tuple(u_final.shape) != (R, n_final):
        raise ValueError(f"u_final must be [{R}, {n_final}]")
    d_fine = torch.empty(R, n_final, dtype=torch.float32, device=dev)
    beta_map = torch.empty(R, dtype=torch.float32, device=dev)
    usage = torch.empty(R, dtype=torch.float32, device=dev)
    nb = lib.nerfart_volsdf_sampler_workspace_bytes(R, n_init, n_up, n_final, max_iter)
    ws = _workspace(nb, dev)
    esc_blob, esc_prec = escalate if (escalate is not None and guard > 0) else (None, 0)
    n_esc = C.c_int(0)
    _check(lib.nerfart_volsdf_fine_sample_guarded2(_dev(surf_blob), int(precision), _dev(esc_blob, name="esc_blob"), int(esc_prec), float(guard), int(late_round),
                                                  _dev(rays_o), _dev(rays_dn), R, None, None, float(near), float(far),
                                                  float(R_bg), float(alpha), float(beta), float(eps), n_init, n_up, n_final, max_iter,
                                                  max_bisect, _dev(lin_table(n_init, dev)), _dev(lin_table(n_up + 2, dev)),
                                                  _dev(lin_table(n_final, dev) if u_final is None else u_final, name="u_final"),
                                                  int(u_final is not None), _dev(d_fine), _dev(beta_map), _dev(usage), C.byref(n_esc), ws.data_ptr(),
                                                  ws.numel(), _stream()), "nerfart_volsdf_fine_sample_guarded2")
    if stats is not None:
        stats["escalated"] = stats.get("escalated", 0) + n_esc.value
        stats["rays"] = stats.get("rays", 0) + R
    return d_fine, beta_map, usage


def sdf_fwd2(surf_blob, pts, direction):
    """Forward sweep of the second-order SDF backward: (value, tangent along `direction`) column pairs; returns the dump
    (uint8) with the activations / tangents (bf16) and softplus' (unorm16) of all 8 layers."""
    M = pts.shape[0]
    dump = torch.empty(int(lib.nerfart_sdf_fwd2_dump_bytes(M)), dtype=torch.uint8, device=pts.device)
    _check(lib.nerfart_sdf_fwd2(_dev(surf_blob), _dev(pts, name="pts"), _dev(direction, name="direction"), M, dump.data_ptr(), _stream()),
           "nerfart_sdf_fwd2")
    return dump


def sdf_bwd2(surf_blob, gbar_h7, gbar_sdf, f2_dump):
    """Reverse sweep: cotangents of h7 [M,256] and sdf [M] -> dump (uint8) of 65535 * (t_l d_l | zbar_l), bf16."""
    M = gbar_sdf.shape[0]
    dump = torch.empty(int(lib.nerfart_sdf_bwd2_dump_bytes(M)), dtype=torch.uint8, device=gbar_sdf.device)
    _check(lib.nerfart_sdf_bwd2(_dev(surf_blob), M, _dev(gbar_h7, name="gbar_h7"), _dev(gbar_sdf, name="gbar_sdf"), f2_dump.data_ptr(),
                                dump.data_ptr(), _stream()), "nerfart_sdf_bwd2")
    return dump


def radiance_fwd_dump(rad_blob, view_tiles: int, pts, view, nabla, h7):
    """Radiance net forward (split-bf16 blob) that also dumps the layer activations (bf16, unit order) for
    radiance_bwd / the weight-gradient GEMMs.  Returns (rgb [M,3], dump uint8)."""
    M = pts.shape[0]
    rgb = torch.empty(M, 3, dtype=torch.float32, device=pts.device)
    dump = torch.empty(int(lib.nerfart_radiance_dump_bytes(M)), dtype=torch.uint8, device=pts.device)
    _check(lib.nerfart_radiance_fwd_dump(_dev(rad_blob), int(view_tiles), _dev(pts, name="pts"), _dev(view, name="view"), M,
                                         _dev(nabla, name="nabla"), _dev(h7, name="h7"), _dev(rgb), dump.data_ptr(), _stream()),
           "nerfart_radiance_fwd_dump")
    return rgb, dump


def radiance_bwd(rad_blob, rgb, g_rgb, fwd_dump):
    """(g_h7 [M,256], g_n [M,3], delta dump uint8) for d loss / d rgb = g_rgb [M,3]."""
    M = rgb.shape[0]
    g_h7 = torch.empty(M, 256, dtype=torch.float32, device=rgb.device)
    g_n = torch.empty(M, 3, dtype=torch.float32, device=rgb.device)
    bwd_dump = torch.empty_like(fwd_dump)
    _check(lib.nerfart_radiance_bwd(_dev(rad_blob), M, _dev(rgb), _dev(g_rgb), fwd_dump.data_ptr(), bwd_dump.data_ptr(), _dev(g_h7),
                                    _dev(g_n), _stream()), "nerfart_radiance_bwd")
    return g_h7, g_n, bwd_dump


def volsdf_composite(d_all, sdf, radiance, alpha: float, beta: float, white_bkgd: bool = False):
    """rgb [R,3], depth [R], acc [R] of nerfart_volsdf_composite for d_all / sdf [R,P], radiance [R,P,3]."""
    R, P = d_all.shape
    dev = d_all.device
    rgb = torch.empty(R, 3, dtype=torch.float32, device=dev)
    depth = torch.empty(R, dtype=torch.float32, device=dev)
    acc = torch.empty(R, dtype=torch.float32, device=dev)
    _check(lib.nerfart_volsdf_composite(R, P, _dev(d_all), _dev(sdf), _dev(radiance), None, float(alpha), float(beta),
                                        int(bool(white_bkgd)), _dev(rgb), _dev(depth), _dev(acc), None, None, None, None, _stream()),
           "nerfart_volsdf_composite")
    return rgb, depth, acc


def volsdf_composite_bwd(d_all, sdf, radiance, alpha: float, beta: float, g_rgb, white_bkgd: bool = False, g_acc=None):
    """Cotangents (g_sdf [R,P], g_radiance [R,P,3], g_alpha_beta [2]) for d loss / d rgb = g_rgb [R,3] (and d loss / d acc
    = g_acc [R], the opacity / mask_volume output, if given)."""
    R, P = d_all.shape
    dev = d_all.device
    g_sdf = torch.empty(R, P, dtype=torch.float32, device=dev)
    g_rad = torch.empty(R, P, 3, dtype=torch.float32, device=dev)
    g_ab = torch.zeros(2, dtype=torch.float32, device=dev)
    _check(lib.nerfart_volsdf_composite_bwd(R, P, _dev(d_all), _dev(sdf), _dev(radiance), float(alpha), float(beta),
                                            int(bool(white_bkgd)), _dev(g_rgb), _dev(g_acc, name="g_acc"), _dev(g_sdf), _dev(g_rad), _dev(g_ab),
                                            _stream()), "nerfart_volsdf_composite_bwd")
    return g_sdf, g_rad, g_ab


def neus_composite_bwd(sdf, rad_mid, s: float, g_rgb, white_bkgd: bool = False, g_acc=None):
    """(g_sdf [R,P], g_rad_mid [R,P-1,3], g_s [1]) for d loss / d rgb = g_rgb [R,3] (and d loss / d acc = g_acc [R])."""
    R, P = sdf.shape
    dev = sdf.device
    g_sdf = torch.empty(R, P, dtype=torch.float32, device=dev)
    g_rad = torch.empty(R, P - 1, 3, dtype=torch.float32, device=dev)
    g_s = torch.zeros(1, dtype=torch.float32, device=dev)
    _check(lib.nerfart_neus_composite_bwd(R, P, _dev(sdf), _dev(rad_mid), float(s), int(bool(white_bkgd)), _dev(g_rgb), _dev(g_acc, name="g_acc"),
                                          _dev(g_sdf), _dev(g_rad), _dev(g_s), _stream()), "nerfart_neus_composite_bwd")
    return g_sdf, g_rad, g_s


# ---- B1 "bwd": the ray-level backward (csrc/render_backward.hip) --------------------------------------------------
RAW_SECTIONS = ("surf_ww", "surf_we", "surf_cs0", "surf_cs17", "surf_w8", "surf_b8", "rad_ww", "rad_cs", "rad_w4", "rad_b4", "rad_wex", "rad_wh7",
                "scalars")
RAW_G_ALPHA, RAW_G_BETA, RAW_G_S, RAW_EIKONAL = 0, 1, 2, 3            # entries of the `scalars` section


def raw_layout():
    """({section: float offset}, total floats) of the raw parameter-gradient buffer (nerfart_pass2_raw_layout)."""
    offs = (C.c_longlong * (len(RAW_SECTIONS) + 1))()
    total = int(lib.nerfart_pass2_raw_layout(offs))
    return {k: int(offs[i]) for i, k in enumerate(RAW_SECTIONS)}, total


def new_raw(device) -> torch.Tensor:
    """A zeroed raw buffer: the entry points below ACCUMULATE into it (one buffer per optimiser step)."""
    return torch.zeros(raw_layout()[1], dtype=torch.float32, device=device)


def volsdf_render_bwd(surf_blob, rad_blob, view_tiles: int, multires: int, rays_o, rays_d, d_all, g_rgb, raw, *, R_bg: float, alpha: float,
                      beta: float, white_bkgd: bool = False, w_eikonal: float = 0.0, eik_group_rays: int = 0, train_radiance: bool = True,
                      g_acc=None, g_n_extra=None, state=None):
    """rgb.backward(g_rgb) + eikonal.backward() of one launch group of VolSDF rays (volsdf.py:759-770) -> accumulated into `raw`.
    rays_d un-normalised; d_all [R, P] from pass 1; state = (sdf [R P], nabla [R P, 3], h7 [R P, 256]) kept from pass 1 or None."""
    R, P = d_all.shape
    need_term(surf_blob, "bf16", "nerfart_volsdf_render_bwd"); need_term(rad_blob, "bf16", "nerfart_volsdf_render_bwd")
    sdf, nab, h7 = state if state is not None else (None, None, None)
    nb = int(lib.nerfart_volsdf_render_bwd_workspace_bytes(R, P, int(state is not None)))
    ws = _workspace(nb, d_all.device)
    _check(lib.nerfart_volsdf_render_bwd(
        _dev(surf_blob), _dev(rad_blob), int(view_tiles), int(multires), _dev(rays_o, name="rays_o"), _dev(rays_d, name="rays_d"), R, P,
        _dev(d_all, name="d_all"), _dev(g_rgb, name="g_rgb"), _dev(g_acc, name="g_acc"), _dev(g_n_extra, name="g_n_extra"),
        _dev(sdf, name="sdf_state"), _dev(nab, name="nabla_state"), _dev(h7, name="h7_state"), float(R_bg), float(alpha), float(beta),
        int(bool(white_bkgd)), float(w_eikonal), int(eik_group_rays or 0), int(bool(train_radiance)), _dev(raw, name="raw"), ws.data_ptr(),
        ws.numel(), _stream()), "nerfart_volsdf_render_bwd")


def neus_render_bwd(surf_blob, rad_blob, view_tiles: int, multires: int, rays_o, rays_d, d_all, g_rgb, raw, *, s: float, white_bkgd: bool = False,
                    w_eikonal: float = 0.0, eik_group_rays: int = 0, train_radiance: bool = False, g_acc=None, state=None):
    """The same for NeuS (neus.py:520-576): state = (sdf [R P], nabla [R P, 3]) at the samples or None."""
    R, P = d_all.shape
    need_term(surf_blob, "bf16", "nerfart_neus_render_bwd"); need_term(rad_blob, "bf16", "nerfart_neus_render_bwd")
    sdf, nab = state if state is not None else (None, None)
    nb = int(lib.nerfart_neus_render_bwd_workspace_bytes(R, P, int(state is not None)))
    ws = _workspace(nb, d_all.device)
    _check(lib.nerfart_neus_render_bwd(
        _dev(surf_blob), _dev(rad_blob), int(view_tiles), int(multires), _dev(rays_o, name="rays_o"), _dev(rays_d, name="rays_d"), R, P,
        _dev(d_all, name="d_all"), _dev(g_rgb, name="g_rgb"), _dev(g_acc, name="g_acc"), _dev(sdf, name="sdf_state"), _dev(nab, name="nabla_state"),
        float(s), int(bool(white_bkgd)), float(w_eikonal), int(eik_group_rays or 0), int(bool(train_radiance)), _dev(raw, name="raw"),
        ws.data_ptr(), ws.numel(), _stream()), "nerfart_neus_render_bwd")


def sdf_param_bwd(surf_blob, multires: int, pts, nbar, raw, sbar=None, hbar7=None):
    """Parameter gradients of sbar . sdf + hbar7 . h7 + nbar . grad_x sdf at pts [M, 3], accumulated into `raw`."""
    M = pts.shape[0]
    need_term(surf_blob, "bf16", "nerfart_sdf_param_bwd")
    nb = int(lib.nerfart_sdf_param_bwd_workspace_bytes(M))
    ws = _workspace(nb, pts.device)
    _check(lib.nerfart_sdf_param_bwd(_dev(surf_blob), int(multires), _dev(pts, name="pts"), M, _dev(sbar, name="sbar"), _dev(hbar7, name="hbar7"),
                                     _dev(nbar, name="nbar"), _dev(raw, name="raw"), ws.data_ptr(), ws.numel(), _stream()), "nerfart_sdf_param_bwd")


def radiance_param_bwd(rad_blob, view_tiles: int, pts, view, nabla, h7, g_rgb, raw, train_radiance: bool = True):
    """(rgb [M,3], g_h7 [M,256], g_n [M,3]); the radiance net's (and the geometry-feature rows') gradients accumulated into `raw`."""
    M = pts.shape[0]
    dev = pts.device
    rgb = torch.empty(M, 3, dtype=torch.float32, device=dev)
    g_h7 = torch.empty(M, 256, dtype=torch.float32, device=dev)
    g_n = torch.empty(M, 3, dtype=torch.float32, device=dev)
    nb = int(lib.nerfart_radiance_param_bwd_workspace_bytes(M))
    ws = _workspace(nb, dev)
    _check(lib.nerfart_radiance_param_bwd(_dev(rad_blob), int(view_tiles), _dev(pts, name="pts"), _dev(view, name="view"), _dev(nabla, name="nabla"),
                                          _dev(h7, name="h7"), M, _dev(g_rgb, name="g_rgb"), _dev(rgb), _dev(g_h7), _dev(g_n), int(bool(train_radiance)),
                                          _dev(raw, name="raw"), ws.data_ptr(), ws.numel(), _stream()), "nerfart_radiance_param_bwd")
    return rgb, g_h7, g_n


def folded_grads_layout(multires: int, multires_view: int):
    """(offsets [29], total): float offsets of (dW_l, db_l) for the SDF net's layers 0..8, then the radiance net's 0..4."""
    offs = (C.c_longlong * 29)()
    total = int(lib.nerfart_folded_grads_layout(int(multires), int(multires_view), offs))
    return [int(o) for o in offs], total


def fold_weight_grads(raw, multires: int, multires_view: int):
    """raw -> (folded [total] fp32, offsets): gradients of the folded weights / biases in the reference's feature order."""
    offs, total = folded_grads_layout(multires, multires_view)
    folded = torch.empty(total, dtype=torch.float32, device=raw.device)
    _check(lib.nerfart_fold_weight_grads(_dev(raw, name="raw"), int(multires), int(multires_view), _dev(folded), _stream()), "nerfart_fold_weight_grads")
    return folded, offs


def weight_norm_bwd(dW, weight_v, weight_g):
    """(g_weight_v [out, in], g_weight_g [out, 1]) of nn.utils.weight_norm for d loss / d W = dW [out, in]."""
    out_f, in_f = weight_v.shape
    g_v = torch.empty_like(weight_v)
    g_g = torch.empty_like(weight_g)
    _check(lib.nerfart_weight_norm_bwd(_dev(dW, name="dW"), _dev(weight_v, name="weight_v"), _dev(weight_g, name="weight_g"), out_f, in_f, _dev(g_v),
                                       _dev(g_g), 0, _stream()), "nerfart_weight_norm_bwd")
    return g_v, g_g


def volsdf_render(surf_blob, rad_blob, view_tiles, rays_o, rays_d, *, near, far, R_bg, alpha, beta, eps=0.1,
                  n_samples=128, n_importance=64, max_upsample_steps=5, max_bisection_steps=10, white_bkgd=False,
                  calc_normal=True, detailed=False, k3_rays_chunk=8192, precision=0, u_final=None, sampler=None, guard: float = 0.0,
                  radiance=None, stats=None, late_round: int = 0):
    """One chunk of rays through nerfart_volsdf_render_staged2_fwd (late_round: the guarded sampler's third rule, hip.volsdf_fine_sample).  Returns a dict of flat [R, ...] tensors.
    u_final [R, n_importance]: uniform random numbers of the final samples (perturb=True); None: deterministic.
    sampler = (surface blob, precision id): Algorithm 1 on its own blob / precision; None: the model's.  guard > 0: the guarded sampler (marginal
    and never-converged rays sampled again on (surf_blob, precision)).  radiance = (radiance blob, precision id): the radiance net of the final
    samples on its own blob / precision; None: (rad_blob, precision).  stats (dict): 'escalated' / 'rays' accumulated."""
    R = rays_o.shape[0]
    dev = rays_o.device
    P = n_samples + n_importance
    if u_final is not None and tuple(u_final.shape) != (R, n_importance):
        raise ValueError(f"u_final must be [{R}, {n_importance}]")
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    out = {"rgb": f(R, 3), "depth_volume": f(R), "mask_volume": f(R)}
    if calc_normal:
        out["normals_volume"] = f(R, 3)
    det = {}
    if detailed:
        det = {"d_vals": f(R, P), "implicit_surface": f(R, P), "implicit_nablas": f(R, P, 3), "radiance": f(R, P, 3),
               "sigma": f(R, P), "p_i": f(R, P - 1), "visibility_weights": f(R, P - 1), "beta_map": f(R), "iter_usage": f(R)}
    nb = lib.nerfart_volsdf_render_workspace_bytes(R, n_samples, n_importance, max_upsample_steps, k3_rays_chunk)
    ws = _workspace(nb, dev)
    g = lambda k: _dev(det.get(k))
    samp_blob, samp_prec = sampler if sampler is not None else (surf_blob, precision)
    rblob, rprec = radiance if radiance is not None else (rad_blob, precision)
    n_esc = C.c_int(0)
    _check(lib.nerfart_volsdf_render_staged2_fwd(
        _dev(surf_blob), int(precision), _dev(rblob, name="rad_blob"), int(rprec), _dev(samp_blob, name="sampler_blob"), int(samp_prec), float(guard),
        int(late_round), int(view_tiles), _dev(rays_o, name="rays_o"), _dev(rays_d, name="rays_d"), R,
        float(near), float(far), float(R_bg), float(alpha), float(beta), float(eps), n_samples, n_importance,
        max_upsample_steps, max_bisection_steps, int(bool(white_bkgd)), k3_rays_chunk,
        _dev(lin_table(n_samples, dev)), _dev(lin_table(4 * n_samples, dev)), _dev(lin_table(4 * n_samples + 2, dev)),
        _dev(lin_table(n_importance, dev) if u_final is None else u_final, name="u_final"), int(u_final is not None),
        _dev(out["rgb"]), _dev(out["depth_volume"]), _dev(out["mask_volume"]), _dev(out.get("normals_volume")),
        g("d_vals"), g("implicit_surface"), g("implicit_nablas"), g("radiance"), g("sigma"), g("p_i"),
        g("visibility_weights"), g("beta_map"), g("iter_usage"), C.byref(n_esc), ws.data_ptr(), ws.numel(), _stream()),
        "nerfart_volsdf_render_staged2_fwd")
    if stats is not None:
        stats["escalated"] = stats.get("escalated", 0) + n_esc.value
        stats["rays"] = stats.get("rays", 0) + R
    out.update(det)
    return out


def volsdf_render_mixed(surf_blob, rad_blob, sampler_blob, sampler_precision: int, view_tiles, rays_o, rays_d, *, near, far, R_bg, alpha, beta, eps=0.1,
                        n_samples=128, n_importance=64, max_upsample_steps=5, max_bisection_steps=10, white_bkgd=False, calc_normal=True,
                        detailed=False, k3_rays_chunk=8192, precision=1, u_final=None, guard: float = 0.0, late_round: int = 0):
    """volsdf_render(..., sampler=(sampler_blob, sampler_precision), guard=guard) restated on the PER-STAGE entry points, in the fused renderer's own order
    (tests: the two must agree bit for bit, every output): the SAMPLER (Algorithm 1: 512 (1 + rounds) SDF queries per ray, no gradient,
    volsdf.py:479) on another blob / precision than the 192 final samples.  The final samples -
    sdf, nabla, radiance, compositing, i.e. every number that reaches a pixel - run at `precision` on (surf_blob, rad_blob); only WHERE the
    64 fine samples sit comes from the cheaper arithmetic.  Same return dict as volsdf_render."""
    R = rays_o.shape[0]
    dev = rays_o.device
    P = n_samples + n_importance
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    dn = normalize_dirs(rays_d)
    d_fine, beta_map, usage = volsdf_fine_sample(sampler_blob, rays_o, dn, near, far, R_bg, alpha, beta, eps, 4 * n_samples, 4 * n_samples, n_importance,
                                                 max_upsample_steps, max_bisection_steps, precision=sampler_precision, u_final=u_final,
                                                 escalate=(surf_blob, precision), guard=guard, late_round=late_round)
    d_coarse = f(R, n_samples)
    _check(lib.nerfart_linspace_depths(_dev(lin_table(n_samples, dev)), n_samples, None, None, float(near), float(far), R, _dev(d_coarse), n_samples,
                                       _stream()), "nerfart_linspace_depths")
    d_all = f(R, P)
    _check(lib.nerfart_sort_concat(R, _dev(d_coarse), n_samples, n_samples, _dev(d_fine), n_importance, n_importance, _dev(d_all), P, _stream()),
           "nerfart_sort_concat")
    sdf, nab, rad = f(R, P), f(R, P, 3), f(R, P, 3)
    ws, nb = nabla_workspace(precision, dev)
    h7 = f(min(k3_rays_chunk, R) * P, 256)
    for c0 in range(0, R, k3_rays_chunk):
        rk = min(k3_rays_chunk, R - c0)
        sl = slice(c0, c0 + rk)
        _check(lib.nerfart_sdf_nabla_fwd_rays(_dev(surf_blob), int(precision), _dev(rays_o[sl]), _dev(dn[sl]), None, _dev(d_all[sl]), rk, P, P, float(R_bg),
                                              _dev(sdf[sl]), _dev(nab[sl]), _dev(h7), _dev(ws, torch.uint8), nb, _stream()), "nerfart_sdf_nabla_fwd_rays")
        _check(lib.nerfart_radiance_fwd_rays(_dev(rad_blob), int(precision), int(view_tiles), _dev(rays_o[sl]), _dev(dn[sl]), None, _dev(d_all[sl]), rk, P, P,
                                             _dev(nab[sl]), _dev(h7), _dev(rad[sl]), _stream()), "nerfart_radiance_fwd_rays")
    out = {"rgb": f(R, 3), "depth_volume": f(R), "mask_volume": f(R)}
    if calc_normal:
        out["normals_volume"] = f(R, 3)
    det = {}
    if detailed:
        det = {"d_vals": d_all, "implicit_surface": sdf, "implicit_nablas": nab, "radiance": rad, "sigma": f(R, P), "p_i": f(R, P - 1),
               "visibility_weights": f(R, P - 1), "beta_map": beta_map, "iter_usage": usage}
    _check(lib.nerfart_volsdf_composite(R, P, _dev(d_all), _dev(sdf), _dev(rad), _dev(nab), float(alpha), float(beta), int(bool(white_bkgd)), _dev(out["rgb"]),
                                        _dev(out["depth_volume"]), _dev(out["mask_volume"]), _dev(out.get("normals_volume")), _dev(det.get("sigma")),
                                        _dev(det.get("p_i")), _dev(det.get("visibility_weights")), _stream()), "nerfart_volsdf_composite")
    out.update(det)
    return out


NEUS_UPSAMPLE_ALGOS = {"official_solution": 0, "direct_use": 1, "direct_more": 2}      # neus.py:242-303


def neus_sample(surf_blob, rays_o, rays_d, *, obj_bounding_radius, n_samples=64, n_importance=64, n_upsample_iters=4, precision=0, u_new=None,
                upsample_algo="official_solution", n_nograd_samples=2048, fixed_s_recp=1 / 64.):
    """The SAMPLER of nerfart_neus_render_algo_fwd on its own, on the stage entry points in the fused renderer's order (near / far, coarse depths,
    SDF queries, up-sampling steps, merges): the P = n_samples + n_importance sorted sample depths [R, P] of every ray, bit-identical to the
    renderer's `d_all` (tests) - what pass 2 of a perturb=True NeuS fine-tune step needs, without rendering a frame to get it (neus.py:240-303 run
    under no_grad: only the depths leave the block)."""
    if upsample_algo not in NEUS_UPSAMPLE_ALGOS:
        raise ValueError(f"upsample_algo must be one of {list(NEUS_UPSAMPLE_ALGOS)}")
    algo = NEUS_UPSAMPLE_ALGOS[upsample_algo]
    if algo == 2 and (4 * int(n_nograd_samples) + max(64, n_importance)) * 4 > 160 * 1024:
        raise NerfartHipError("upsample_algo 'direct_more' keeps 4 x N_nograd_samples floats per ray in the 160 KiB LDS: N_nograd_samples must be <= 10,200")
    R, dev = rays_o.shape[0], rays_o.device
    P = n_samples + n_importance
    if u_new is not None and tuple(u_new.shape) != (R, n_importance):
        raise ValueError(f"u_new must be [{R}, {n_importance}]")
    f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
    if R == 0:
        return f(0, P)
    dn = normalize_dirs(rays_d)
    near, far = f(R), f(R)
    _check(lib.nerfart_near_far_from_sphere(_dev(rays_o, name="rays_o"), _dev(dn), R, float(obj_bounding_radius), _dev(near), _dev(far), _stream()),
           "nerfart_near_far_from_sphere")
    d, sd = f(R, P), f(R, P)

    def depths(table, n, out, stride):
        _check(lib.nerfart_linspace_depths(_dev(table), n, _dev(near), _dev(far), 0.0, 0.0, R, _dev(out), stride, _stream()), "nerfart_linspace_depths")

    def query(depth, n, stride, out, out_stride):
        _check(lib.nerfart_sdf_fwd_rays(_dev(surf_blob), int(precision), _dev(rays_o), _dev(dn), None, _dev(depth), R, n, stride, 0.0, _dev(out), out_stride,
                                        _stream()), "nerfart_sdf_fwd_rays")

    def merge(n, d_new, s_new, n_new):
        _check(lib.nerfart_merge_sorted_pairs(R, n, P, n_new, _dev(d), _dev(sd), _dev(d_new), _dev(s_new), _stream()), "nerfart_merge_sorted_pairs")
    depths(lin_table(n_samples, dev), n_samples, d, P)
    query(d, n_samples, P, sd, P)
    n = n_samples
    if algo == 0:
        n_new = n_importance // n_upsample_iters
        table = lin_table(n_new, dev) if u_new is None else u_new.contiguous()
        d_new, s_new = f(R, n_new), f(R, n_new)
        for i in range(n_upsample_iters):
            u_ptr = table.data_ptr() + (4 * i * n_new if u_new is not None else 0)
            _check(lib.nerfart_neus_upsample_step(R, n, P, n_new, 64.0 * (1 << i), _dev(d), _dev(sd), u_ptr, n_importance if u_new is not None else 0,
                                                  _dev(d_new), _stream()), "nerfart_neus_upsample_step")
            query(d_new, n_new, n_new, s_new, n_new)
            merge(n, d_new, s_new, n_new)
            n += n_new
    else:
        bins_d, bins_s, n_bins, cap = d, sd, n_samples, P
        if algo == 2:
            bins_d, bins_s, n_bins, cap = f(R, n_nograd_samples), f(R, n_nograd_samples), int(n_nograd_samples), int(n_nograd_samples)
            depths(lin_table(n_bins, dev), n_bins, bins_d, n_bins)
            query(bins_d, n_bins, n_bins, bins_s, n_bins)
        table = lin_table(n_importance, dev) if u_new is None else u_new.contiguous()
        d_new, s_new = f(R, n_importance), f(R, n_importance)
        _check(lib.nerfart_neus_direct_upsample_step(R, n_bins, cap, n_importance, 1.0 / float(fixed_s_recp), _dev(bins_d), _dev(bins_s), _dev(table),
                                                     n_importance if u_new is not None else 0, _dev(d_new), _stream()), "nerfart_neus_direct_upsample_step")
        query(d_new, n_importance, n_importance, s_new, n_importance)
        merge(n, d_new, s_new, n_importance)
    return d


def neus_render(surf_blob, rad_blob, view_tiles, rays_o, rays_d, *, obj_bounding_radius, s, n_samples=64, n_importance=64,
                n_upsample_iters=4, white_bkgd=False, calc_normal=True, detailed=False, k3_rays_chunk=8192, precision=0,
                u_new=None, upsample_algo="official_solution", n_nograd_samples=2048, fixed_s_recp=1 / 64.):
    """One chunk of rays through nerfart_neus_render_algo_fwd.  u_new [R, n_importance]: uniform random numbers of the
    up-sampling (perturb=True; 'official_solution': round i takes columns i * n_new ..; 'direct_use' / 'direct_more': one inversion
    of all n_importance); None: deterministic."""
    R = rays_o.shape[0]
    dev = rays_o.device
    P = n_samples + n_importance
    if upsample_algo not in NEUS_UPSAMPLE_ALGOS:
        raise ValueError(f"upsample_algo must be one of {list(NEUS_UPSAMPLE_ALGOS)}")
    algo = NEUS_UPSAMPLE_ALGOS[upsample_algo]
    if u_new is not None and tuple(u_new.shape) != (R, n_importance):
        raise ValueError(f"u_new must be [{R}, {n_importance}]")
    f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
    out = {"rgb": f(R, 3), "depth_volume": f(R), "mask_volume": f(R)}
    if calc_normal:
        out["normals_volume"] = f(R, 3)
    det = {}
    if detailed:
        det = {"d_all": f(R, P), "implicit_surface": f(R, P), "implicit_nablas": f(R, P, 3), "radiance": f(R, P - 1, 3),
               "cdf": f(R, P), "alpha": f(R, P - 1), "visibility_weights": f(R, P - 1), "d_final": f(R, P - 1)}
    nb = lib.nerfart_neus_render_algo_workspace_bytes(R, n_samples, n_importance, k3_rays_chunk, algo, int(n_nograd_samples))
    ws = _workspace(nb, dev)
    g = lambda k: _dev(det.get(k))
    n_new = n_importance // n_upsample_iters if algo == 0 else n_importance
    _check(lib.nerfart_neus_render_algo_fwd(
        _dev(surf_blob), _dev(rad_blob), int(precision), int(view_tiles), _dev(rays_o, name="rays_o"), _dev(rays_d, name="rays_d"), R,
        float(obj_bounding_radius), float(s), n_samples, n_importance, n_upsample_iters, algo, int(n_nograd_samples), float(fixed_s_recp),
        int(bool(white_bkgd)), k3_rays_chunk,
        _dev(lin_table(n_samples, dev)), _dev(lin_table(int(n_nograd_samples), dev) if algo == 2 else None),
        _dev(lin_table(n_new, dev) if u_new is None else u_new, name="u_new"),
        int(u_new is not None), _dev(out["rgb"]), _dev(out["depth_volume"]), _dev(out["mask_volume"]), _dev(out.get("normals_volume")),
        g("d_all"), g("implicit_surface"), g("implicit_nablas"), g("radiance"), g("cdf"), g("alpha"),
        g("visibility_weights"), g("d_final"), ws.data_ptr(), ws.numel(), _stream()), "nerfart_neus_render_algo_fwd")
    out.update(det)
    return out
