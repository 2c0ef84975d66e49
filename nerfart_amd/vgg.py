"""VGG16 perceptual loss of the fine-tune objective (reference criteria/perp_loss.py:9-57, weight
`w_perceptual` = 2.0 in configs/volsdf_fangzhou_vangogh.yaml:82, used at volsdf.py:898-899).

The reference runs torchvision's `vgg16(pretrained=True).features` in four slices ([:4], [4:9], [9:16], [16:23]) on the
ImageNet-normalised, bilinearly 224 x 224-resized prediction and target and takes the L1 distance of the THIRD slice's
output only (relu3_3; `if i == 2`, perp_loss.py:51) - the fourth slice is computed and dropped, so it is not built here.
On the GPU the conv stack, the L1 and the backward pass to the prediction's pixels run on the hand-written kernels of
csrc/vgg_conv.hip (implicit-GEMM convolutions on the matrix cores, fp32 operands like the reference's torchvision net, behind
`nerfart_vgg16_l1_fwd / _bwd`); the ImageNet normalisation + bilinear resize in front is one `nerfart_resample_fwd` gather.
The torch formulation below (im2col via `F.unfold` + matmul, prediction and target as one batch of two) is the CPU path.
`native=False` at construction forces the torch formulation on the GPU too (the cross-check; it warns once when it runs).

Weights: pass torchvision's `vgg16` state dict (`features.N.weight / bias`; N = 0, 2, 5, 7, 10, 12, 14 are read).  No
ImageNet checkpoint exists offline, so the default is torchvision's own initialiser (seeded).  The arithmetic - both
formulations - is pinned against the reference's own criteria/perp_loss.py run on a torchvision-shaped net with these weights
(tests/golden/make_golden_style.py -> tests/test_style_golden.py, tests/test_gpu_style_golden.py).
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

    def __init__(self, state_dict=None, resize: bool = True, seed: int = 0, native: bool = None):
        super().__init__()
        # native=False: the torch formulation on the GPU too - the cross-check the kernels are tested against; an explicit constructor
        # argument only (no environment switch), and it warns once when it runs (criteria._warn_library_path)
        self.native = True if native is None else bool(native)
        self.net = VGG16Features(state_dict, seed)
        self.register_buffer("mean", torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor(IMAGENET_STD).view(1, 3, 1, 1))
        self.resize = resize

    # ---- hand-written path (GPU) ------------------------------------------------------------------------------
    def packed(self):
        """The conv weights as the kernels read them (section order of csrc/vgg_conv.hip), re-packed when they change."""
        import ctypes as C
        from . import hip
        key = tuple((p.data_ptr(), p._version) for p in self.net.parameters())
        if getattr(self, "_blob_key", None) != key:
            total = int(hip.lib.nerfart_vgg16_blob_layout(None))
            dev = self.mean.device
            blob = torch.empty(total, dtype=torch.uint8, device=dev)
            # nerfart_vgg16_pack: forward [Cout, 9 Cin], flipped-transposed backward [Cin, 9 Cout] and bias sections, on the device
            ws = [self.net.features[str(idx)].weight.detach().to(device=dev, dtype=torch.float32).contiguous() for idx, _, _, _ in _CONVS]
            bs = [self.net.features[str(idx)].bias.detach().to(device=dev, dtype=torch.float32).contiguous() for idx, _, _, _ in _CONVS]
            wt = (C.c_void_p * len(ws))(*[t.data_ptr() for t in ws])
            bt = (C.c_void_p * len(bs))(*[t.data_ptr() for t in bs])
            with torch.cuda.device(dev):
                rc = hip.lib.nerfart_vgg16_pack(wt, bt, C.c_void_p(blob.data_ptr()), total, C.c_void_p(torch.cuda.current_stream().cuda_stream))
            if rc != 0:
                raise RuntimeError("nerfart_vgg16_pack: " + hip.lib.nerfart_last_error().decode())
            self._blob, self._blob_key = blob, key
        return self._blob

    def _native(self, input, target):
        from . import style_native as sn
        dev = input.device
        a = torch.stack([1.0 / self.std.reshape(3), -self.mean.reshape(3) / self.std.reshape(3)]).float().contiguous().to(dev)
        H, W = (224, 224) if self.resize else input.shape[-2:]
        x = sn.resample(input, (H, W), mode="bilinear", affine=a)              # (x - mean) / std commutes with the interpolation
        with torch.no_grad():
            y = sn.resample(target.to(input.dtype), (H, W), mode="bilinear", affine=a)
        return _VGGL1.apply(x, y, self.packed())

    def forward(self, input, target):
        if input.shape[1] != 3:
            input, target = input.repeat(1, 3, 1, 1), target.repeat(1, 3, 1, 1)
        if input.is_cuda and input.shape[0] == 1 and self.native:
            H, W = (224, 224) if self.resize else input.shape[-2:]
            if H % 4 == 0 and W % 4 == 0 and (H * W // 16) % 64 == 0:          # the kernels' tile geometry (always true with resize=True)
                return self._native(input, target)
        if input.is_cuda:                                  # never silently: the torch formulation (library kernels) is running on the GPU
            from .criteria import _warn_library_path
            _warn_library_path("VGGPerceptualLoss: torch formulation on the GPU (native=False, a batch, or a frame size outside the kernels' "
                               "tile geometry) - the cross-check, not the hand-written convolutions")
        xy = torch.cat([input, target.to(input.dtype)], dim=0)
        xy = (xy - self.mean) / self.std
        if self.resize:
            xy = F.interpolate(xy, mode="bilinear", size=(224, 224), align_corners=False)
        f = self.net(xy)
        n = input.shape[0]
        return F.l1_loss(f[:n], f[n:])


class _VGGL1(torch.autograd.Function):
    """mean |relu3_3(x) - relu3_3(y)| on the hand-written kernels; gradient w.r.t. x (the prediction) only."""

    @staticmethod
    def forward(ctx, x, y, blob):
        import ctypes as C
        from . import hip
        H, W = x.shape[-2:]
        img2 = torch.cat([x.detach(), y.detach()], dim=0).float().contiguous()
        keep = bool(ctx.needs_input_grad[0])
        nbytes = hip.lib.nerfart_vgg16_workspace_bytes(H, W, int(keep))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = hip.lib.nerfart_vgg16_l1_fwd(C.c_void_p(blob.data_ptr()), blob.numel() * blob.element_size(), C.c_void_p(img2.data_ptr()), H, W, C.c_void_p(loss.data_ptr()), int(keep),
                                          C.c_void_p(ws.data_ptr()), nbytes, st)
        if rc != 0:
            raise RuntimeError("nerfart_vgg16_l1_fwd: " + hip.lib.nerfart_last_error().decode())
        ctx.keep, ctx.ws, ctx.blob, ctx.hw, ctx.dtype = keep, ws if keep else None, blob, (H, W), x.dtype
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        import ctypes as C
        from . import hip
        if not ctx.keep:
            return None, None, None
        H, W = ctx.hw
        up = g.detach().float().reshape(1).contiguous()
        gi = torch.empty(1, 3, H, W, dtype=torch.float32, device=up.device)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = hip.lib.nerfart_vgg16_l1_bwd(C.c_void_p(ctx.blob.data_ptr()), ctx.blob.numel() * ctx.blob.element_size(), H, W, C.c_void_p(up.data_ptr()), C.c_void_p(gi.data_ptr()),
                                          C.c_void_p(ctx.ws.data_ptr()), ctx.ws.numel(), st)
        if rc != 0:
            raise RuntimeError("nerfart_vgg16_l1_bwd: " + hip.lib.nerfart_last_error().decode())
        ctx.ws = None
        return gi.to(ctx.dtype), None, None
