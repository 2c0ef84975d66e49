"""CPU side of the native CLIP image encoder (rows a23 / B4): the weight blob's section table (C side) and the packer
(Python side) agree - every section round-trips; every matrix is stored once (the sections that held the transposed copies in
round 2 are empty: the backward GEMMs read the forward matrices in place).  No compute (no GPU here)."""
import numpy as np
import torch


def test_blob_layout_and_packing_round_trip():
    from nerfart_amd import clip_native, clip_vit
    offs, total = clip_native.blob_layout()
    assert len(offs) == clip_native.N_SECTIONS + 1 and offs[0] == 0 and offs[-1] == total
    empty = {1, 98} | {2 + 8 * l + j for l in range(12) for j in (1, 3, 5, 7)}
    assert all((b == a if i in empty else b > a) and a % 256 == 0 for i, (a, b) in enumerate(zip(offs[:-1], offs[1:])))
    # fp16 matrices ONCE + fp32 vectors: 2 B x 87.4 M matrix entries + the vectors
    model = clip_vit.build_clip("cpu", seed=0)
    n_mat = sum(p.numel() for n, p in model.visual.named_parameters() if p.dim() >= 2 and "positional" not in n)
    n_vec = sum(p.numel() for n, p in model.visual.named_parameters() if p.dim() < 2 or "positional" in n)
    assert 2 * n_mat + 4 * n_vec <= total < 2 * n_mat + 4 * n_vec + 256 * clip_native.N_SECTIONS
    sd = {"visual." + k: v for k, v in model.visual.state_dict().items()}
    blob = clip_native.pack_visual(sd, "cpu")
    assert blob.numel() == total

    def sec(i, dtype, shape):
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        return blob[offs[i]: offs[i] + n].view(dtype).reshape(shape)
    w = sd["visual.transformer.resblocks.7.mlp.c_proj.weight"].half()
    assert torch.equal(sec(2 + 8 * 7 + 6, torch.float16, (768, 3072)), w)
    assert torch.equal(sec(0, torch.float16, (768, 3072)), sd["visual.conv1.weight"].reshape(768, -1).half())
    assert torch.equal(sec(99, torch.float16, (768, 512)), sd["visual.proj"].half())
    assert torch.equal(sec(101, torch.float32, (50, 768)), sd["visual.positional_embedding"].float())
    assert torch.equal(sec(104 + 8 * 3 + 2, torch.float32, (2304,)), sd["visual.transformer.resblocks.3.attn.in_proj_bias"].float())
    assert torch.equal(sec(201, torch.float32, (768,)), sd["visual.ln_post.bias"].float())
    from nerfart_amd import hip
    assert hip.lib.nerfart_clip_vitb32_workspace_bytes(16, 1) > hip.lib.nerfart_clip_vitb32_workspace_bytes(16, 0) > 0
