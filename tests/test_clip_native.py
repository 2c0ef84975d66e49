"""CPU side of the native CLIP image encoder (rows a23 / B4): the weight blob's section table and the packer's tensor table (both C side)
agree with the tower's parameters; every matrix is stored once (the sections that held the transposed copies in
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
    # the packer is the library's since round 5 (nerfart_clip_vitb32_pack: needs the GPU - its VALUES are held byte for byte to the torch statement
    # of the blob in tests/test_gpu_pack.py); on the CPU: the library names the tensors it takes, and they are exactly the tower's parameters
    names = clip_native.tensor_names()
    assert len(names) == 152 == len(sd) and len({n for n, _ in names}) == 152
    assert all(("visual." + n) in sd and sd["visual." + n].numel() == k for n, k in names)
    n_half = sum(k for n, k in names if sd["visual." + n].dim() >= 2 and "positional" not in n)
    assert n_half == n_mat and sum(k for _, k in names) == n_mat + n_vec
    with __import__("pytest").raises(Exception):
        clip_native.pack_visual(sd, "cpu")                       # no CPU path
    from nerfart_amd import hip
    assert hip.lib.nerfart_clip_vitb32_workspace_bytes(16, 1) > hip.lib.nerfart_clip_vitb32_workspace_bytes(16, 0) > 0
