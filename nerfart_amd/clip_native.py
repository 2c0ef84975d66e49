"""CLIP ViT-B/32 image encoder on the hand-written gfx950 kernels (rows a23 / B4): `encode_image` forward and its backward
w.r.t. the image through the C ABI `nerfart_clip_vitb32_image_fwd` / `_bwd` (csrc/clip_vit.hip, include/nerfart_hip.h).

    enc = NativeImageEncoder(clip_model)          # clip_vit.CLIP (or anything with OpenAI's `visual.*` state-dict names)
    feat = enc(images)                            # [B, 3, 224, 224] normalised -> [B, 512] fp32, differentiable in `images`

This replaces `model.encode_image(images)` at the reference's call sites (criteria/clip_loss.py:204-216,
contrastive_loss.py:110-114, patchnce_loss.py:124-128).  The text encoder stays on the torch module: its outputs are
constants of a run and are cached (criteria.ClipFeatures), SURVEY.md 8b B4.  Weights are frozen (requires_grad False in the
reference's losses too): the backward pass produces the pixel gradient only.
"""
import ctypes as C

import torch

from . import hip

N_SECTIONS = 202
WIDTH, LAYERS, PATCH, RES, OUT = 768, 12, 32, 224, 512


def blob_layout():
    offs = (C.c_longlong * (N_SECTIONS + 1))()
    total = hip.lib.nerfart_clip_vitb32_blob_layout(C.cast(offs, C.c_void_p))
    return list(offs), int(total)


def tensor_names():
    """[(`visual.`-relative state-dict name, element count)] in the order nerfart_clip_vitb32_pack takes its tensors - read from the library."""
    out = []
    buf = C.create_string_buffer(96)
    for i in range(hip.lib.nerfart_clip_vitb32_n_tensors()):
        n = int(hip.lib.nerfart_clip_vitb32_tensor_name(i, buf, len(buf)))
        out.append((buf.value.decode(), n))
    return out


def pack_visual(state, device) -> torch.Tensor:
    """`visual.*` entries of a CLIP state dict -> the weight blob (uint8 tensor on `device`) through nerfart_clip_vitb32_pack: fp16 matrices
    (each stored once), fp32 vectors, in the section order of csrc/clip_vit.hip.  The library names the tensors it wants."""
    _, total = blob_layout()

    def g(name):
        return state["visual." + name].detach()

    if g("conv1.weight").shape != (WIDTH, 3, PATCH, PATCH) or g("positional_embedding").shape != (50, WIDTH) or g("proj").shape != (WIDTH, OUT):
        raise ValueError("clip_native: not a ViT-B/32 visual tower (conv1 768x3x32x32, 50 positions, proj 768x512)")
    tensors = []
    for name, n in tensor_names():
        t = g(name).to(device=device, dtype=torch.float32).contiguous()
        if t.numel() != n:
            raise ValueError(f"clip_native: visual.{name} has {t.numel()} elements, the ViT-B/32 tower has {n}")
        tensors.append(t)
    blob = torch.empty(total, dtype=torch.uint8, device=device)
    table = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    with torch.cuda.device(blob.device):
        _check(hip.lib.nerfart_clip_vitb32_pack(table, _ptr(blob), total, _stream()), "nerfart_clip_vitb32_pack")
    return blob


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: {hip.lib.nerfart_last_error().decode()}")


def image_fwd(blob, images, keep: bool):
    """images [B, 3, 224, 224] fp32 contiguous on the GPU -> (features [B, 512] fp32, workspace)."""
    B = images.shape[0]
    nbytes = hip.lib.nerfart_clip_vitb32_workspace_bytes(B, int(keep))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=images.device)
    feat = torch.empty(B, OUT, dtype=torch.float32, device=images.device)
    _check(hip.lib.nerfart_clip_vitb32_image_fwd(_ptr(blob), blob.numel() * blob.element_size(), _ptr(images), B, _ptr(feat), int(keep), _ptr(ws), nbytes, _stream()),
           "nerfart_clip_vitb32_image_fwd")
    return feat, ws


def image_bwd(blob, ws, B, g_feat):
    g_img = torch.empty(B, 3, RES, RES, dtype=torch.float32, device=g_feat.device)
    _check(hip.lib.nerfart_clip_vitb32_image_bwd(_ptr(blob), blob.numel() * blob.element_size(), B, _ptr(g_feat), _ptr(g_img), _ptr(ws), ws.numel(), _stream()),
           "nerfart_clip_vitb32_image_bwd")
    return g_img


class _Encode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, images, blob):
        x = images.detach().float().contiguous()
        keep = bool(ctx.needs_input_grad[0])
        feat, ws = image_fwd(blob, x, keep)
        if keep:
            ctx.blob, ctx.ws, ctx.B, ctx.in_dtype = blob, ws, x.shape[0], images.dtype
        ctx.keep = keep
        return feat

    @staticmethod
    def backward(ctx, g_feat):
        if not ctx.keep:
            return None, None
        g = image_bwd(ctx.blob, ctx.ws, ctx.B, g_feat.detach().float().contiguous())
        ctx.ws = None
        return g.to(ctx.in_dtype), None


def gemm_f16_nt(a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """a [M, K] fp16 . w [N, K]^T fp16 -> [M, N] fp32 on the encoder's GEMM kernel (tests; M, N, K multiples of 64)."""
    M, K = a.shape
    N = w.shape[0]
    c = torch.empty(M, N, dtype=torch.float32, device=a.device)
    _check(hip.lib.nerfart_gemm_f16_nt(_ptr(a.contiguous()), _ptr(w.contiguous()), M, N, K, _ptr(c), _stream()), "nerfart_gemm_f16_nt")
    return c


def gemm_f16_nn(a: torch.Tensor, wt: torch.Tensor) -> torch.Tensor:
    """a [M, K] fp16 . wt [K, N] fp16 -> [M, N] fp32: the same kernel with the second operand's reduction index as its ROW (what
    the backward GEMMs do with the forward weight matrices; tests)."""
    M, K = a.shape
    N = wt.shape[1]
    c = torch.empty(M, N, dtype=torch.float32, device=a.device)
    _check(hip.lib.nerfart_gemm_f16_nn(_ptr(a.contiguous()), _ptr(wt.contiguous()), M, N, K, _ptr(c), _stream()), "nerfart_gemm_f16_nn")
    return c


class NativeImageEncoder:
    """Callable stand-in for `model.encode_image`; the blob is re-packed when the model's visual parameters change."""

    def __init__(self, clip_model):
        self.model = clip_model
        self._blob, self._key = None, None

    def _params_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.model.visual.parameters())

    def blob(self):
        key = self._params_key()
        if self._blob is None or key != self._key:
            dev = next(self.model.visual.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("clip_native: the CLIP model must be on the GPU (there is no CPU path)")
            self._blob = pack_visual({"visual." + k: v for k, v in self.model.visual.state_dict().items()}, dev)
            self._key = key
        return self._blob

    def __call__(self, images):
        if images.dim() != 4 or tuple(images.shape[1:]) != (3, RES, RES):
            raise ValueError(f"clip_native: images must be [B, 3, {RES}, {RES}], got {tuple(images.shape)}")
        if not images.is_cuda:
            raise RuntimeError("clip_native: images must be on the GPU")
        return _Encode.apply(images, self.blob())
