"""Weight packing for the chained-MLP HIP kernels (csrc/mlp_chain.hip).

The kernels keep activations in MFMA C-layout registers: lane (g, i) of a wave holds, for tile
t and register r, feature *slot* 16t + 4g + r of its column.  A layer is
``out^T = W . h^T`` evaluated k-outer: for every 16-wide k tile, for every 16-wide output
tile T, four ``v_mfma_f32_16x16x4_f32`` whose A fragment for k-step r is
``A[lane=(g,i)] = W[16T + i][16t + 4g + r]``.  This module lays the folded
(``w = g * v / ||v||``, reference models/base.py:226-227) weights out in exactly that order so
that the kernel's weight stream is a linear copy and each lane reads its four k-steps with one
``ds_read_b128``:

    blob  = int32 header[512] | chunk 0 | chunk 1 | ... | aux
    chunk = 1 or 2 k tiles;  k tile = float[T=16][lane=64][r=4]  (16 KiB)

Slots are the reference's feature indices except where the kernel is free to choose:
  * the 39 positional-encoding features occupy 48 slots ordered per lane group
    (``enc_slot_feature``), because lane group g computes the sin/cos of coordinate g;
  * the radiance net's first-layer inputs are ordered [feat(256) | x, view, normal] because
    the 256 geometry features arrive as 16 full tiles.

Nothing here is numerics: packing is a gather (``packed = flat_weights[index]``).
"""
from __future__ import annotations

import numpy as np
import torch

MAGIC = 0x4E414631
HDR_INTS = 512
HDR_OFFS = 16
KT = 16 * 64 * 4          # floats per k tile
PROG_SURFACE = 1
PROG_RADIANCE = 2

SURF_AUX_ROW = 2048
SURF_AUX_B8 = 2304
SURF_AUX_FLOATS = 2308
RAD_AUX_ROWS = 1280
RAD_AUX_BF = 2048
RAD_AUX_FLOATS = 2052


def enc_slot_feature(slot: int, multires: int = 6) -> int:
    """Slot (0..47) of the SDF-net encoding -> reference feature index (0..38) or -1 (padding).

    Reference order (models/base.py:53-61): [x(3), sin(2^0 x)(3), cos(2^0 x)(3), sin(2^1 x)(3), ...].
    Slot order: lane group g < 3 owns coordinate g: m = 0 raw, m = 1+2k sin(2^k .), m = 2+2k cos(2^k .)
    for k < 5, m = 11 pad; lane group 3 owns the last band k = 5: m = 2i sin(32 x_i), m = 2i+1 cos.
    """
    assert multires == 6, "the gfx950 kernels are built for embed_multires = 6 (39 features)"
    t, rem = divmod(slot, 16)
    g, r = divmod(rem, 4)
    m = 4 * t + r
    if g < 3:
        if m == 0:
            return g
        if m == 11:
            return -1
        k, is_cos = divmod(m - 1, 2)
        return 3 + 6 * k + 3 * is_cos + g
    if m < 6:
        i, is_cos = divmod(m, 2)
        return 3 + 6 * 5 + 3 * is_cos + i
    return -1


ENC_SLOTS = np.array([enc_slot_feature(s) for s in range(48)], dtype=np.int64)


class _Flat:
    """Flat concatenation of the source tensors (folded weights, biases) plus one trailing zero."""

    def __init__(self):
        self.names, self.base, self.shape, self.n = [], {}, {}, 0

    def add(self, name, shape):
        self.names.append(name)
        self.base[name] = self.n
        self.shape[name] = tuple(shape)
        self.n += int(np.prod(shape))

    @property
    def zero(self):
        return self.n

    def mat_index(self, name, rows, cols):
        """rows[R], cols[C] (int, -1 = zero) -> [R, C] flat indices."""
        R, C = self.shape[name]
        idx = self.base[name] + rows[:, None] * C + cols[None, :]
        idx = np.where((rows[:, None] < 0) | (cols[None, :] < 0), self.zero, idx)
        return idx

    def vec_index(self, name, sel):
        idx = self.base[name] + sel
        return np.where(sel < 0, self.zero, idx)


def _ktile_index(midx, t):
    """midx [256 out slots, K in slots] flat indices -> k tile t as [T=16][lane=64][r=4]."""
    out = np.empty((16, 64, 4), dtype=np.int64)
    lane = np.arange(64)
    g, i = lane // 16, lane % 16
    for T in range(16):
        for r in range(4):
            out[T, :, r] = midx[16 * T + i, 16 * t + 4 * g + r]
    return out.reshape(-1)


def _ktile_index_n(midx, t, ntiles):
    """midx [16 ntiles out slots, K in slots] -> k tile t as [T = ntiles][lane = 64][r = 4] (the reverse sweep's 3-tile tails)."""
    out = np.empty((ntiles, 64, 4), dtype=np.int64)
    lane = np.arange(64)
    g, i = lane // 16, lane % 16
    for T in range(ntiles):
        for r in range(4):
            out[T, :, r] = midx[16 * T + i, 16 * t + 4 * g + r]
    return out.reshape(-1)


def _layer_chunks(midx, nt_base, nt_extra):
    """Chunk sequence of one layer: ceil(nt_base/2) chunks of base k tiles, then ceil(nt_extra/2)
    chunks of extra k tiles (must match run_layer in mlp_chain.hip)."""
    chunks = []
    for c in range((nt_base + 1) // 2):
        tiles = [2 * c] + ([2 * c + 1] if 2 * c + 1 < nt_base else [])
        chunks.append(np.concatenate([_ktile_index(midx, t) for t in tiles]))
    for c in range((nt_extra + 1) // 2):
        tiles = [nt_base + 2 * c] + ([nt_base + 2 * c + 1] if 2 * c + 1 < nt_extra else [])
        chunks.append(np.concatenate([_ktile_index(midx, t) for t in tiles]))
    return chunks


def _pad(a, n, fill=-1):
    out = np.full(n, fill, dtype=np.int64)
    out[: len(a)] = a
    return out


class PackPlan:
    """Gather plan for one program: ``index`` into the flat source vector, header words.  ``chunks_rev`` (fp32 surface program):
    the transposed-weight chunks of the reverse sweep, consumed right after the forward ones by k_sdf_grad (header word 6 = the
    number of chunks of both sweeps; word 2 stays the forward count that k_sdf_only / k_sdf_nabla run)."""

    def __init__(self, prog, flat, chunks, aux, chunks_rev=()):
        self.prog = prog
        self.flat = flat
        chunks = list(chunks) + list(chunks_rev)
        offs = [HDR_INTS]
        for c in chunks:
            offs.append(offs[-1] + len(c))
        self.nc = len(chunks) - len(chunks_rev)
        self.nc_all = len(chunks)
        assert self.nc_all + 1 <= 128, "chunk table is 128 entries in the kernel"
        self.aux_off = offs[-1]
        self.index = np.concatenate(chunks + [aux])
        self.total = HDR_INTS + len(self.index)
        hdr = np.zeros(HDR_INTS, dtype=np.int32)
        hdr[0], hdr[1], hdr[2], hdr[3], hdr[4], hdr[5] = MAGIC, prog, self.nc, self.total, self.aux_off, len(aux)
        hdr[6] = self.nc_all
        hdr[HDR_OFFS: HDR_OFFS + self.nc_all + 1] = offs
        self.header = hdr
        self._index_t = {}

    def pack(self, tensors: dict) -> torch.Tensor:
        """tensors: name -> tensor (any device, fp32).  Returns the fp32 blob on that device."""
        dev = tensors[self.flat.names[0]].device
        parts = []
        for n in self.flat.names:
            t = tensors[n]
            assert tuple(t.shape) == self.flat.shape[n], (n, tuple(t.shape), self.flat.shape[n])
            parts.append(t.detach().reshape(-1).to(torch.float32))
        parts.append(torch.zeros(1, dtype=torch.float32, device=dev))
        src = torch.cat(parts)
        key = str(dev)
        if key not in self._index_t:
            self._index_t[key] = torch.from_numpy(self.index).to(dev)
        body = src[self._index_t[key]]
        hdr = torch.from_numpy(self.header.copy()).view(torch.float32).to(dev)
        return torch.cat([hdr, body]).contiguous()


def surface_plan(W: int = 256, D: int = 8, skips=(4,), multires: int = 6, W_geo_feat: int = 256) -> PackPlan:
    """SDF net (reference ImplicitSurface, models/base.py:131-263) -> program for k_sdf_only / k_sdf_nabla."""
    if not (W == 256 and D == 8 and tuple(skips) == (4,) and multires == 6 and W_geo_feat == 256):
        raise NotImplementedError("gfx950 surface kernels are built for W=256, D=8, skips=[4], embed_multires=6, "
                                  "W_geo_feat=256 (the four reference configs)")
    enc = 39
    flat = _Flat()
    dims = []
    for l in range(D + 1):
        out_dim = (1 + W_geo_feat) if l == D else (W - enc if (l + 1) in skips else W)
        in_dim = enc if l == 0 else W
        dims.append((out_dim, in_dim))
        flat.add(f"w{l}", (out_dim, in_dim))
        flat.add(f"b{l}", (out_dim,))
    chunks = []
    ar = np.arange
    for l in range(D):
        out_dim, in_dim = dims[l]
        rows = _pad(ar(out_dim), 256)
        if l == 0:
            cols, nb, ne = ENC_SLOTS, 3, 0
        elif l in skips:
            hw = W - enc                                           # 217 -> 14 tiles
            cols = np.concatenate([_pad(ar(hw), 224), np.where(ENC_SLOTS >= 0, hw + ENC_SLOTS, -1)])
            nb, ne = 16, 1
        else:
            cols, nb, ne = _pad(ar(in_dim), 256), 16, 0
        midx = flat.mat_index(f"w{l}", rows, cols)
        chunks += _layer_chunks(midx, nb, ne)
    aux = []
    for l in range(D):
        aux.append(flat.vec_index(f"b{l}", _pad(ar(dims[l][0]), 256)))
    aux.append(flat.mat_index(f"w{D}", np.array([0]), ar(256)).reshape(-1))      # sdf row
    aux.append(flat.vec_index(f"b{D}", _pad(np.array([0]), 4)))
    aux = np.concatenate(aux)
    assert len(aux) == SURF_AUX_FLOATS
    # ---- reverse sweep (k_sdf_grad: d sdf / d a_{l-1} = W_l^T (d sdf / d a_l * softplus'(z_l)), l = 7 .. 0) -----------------------
    # k slots = layer l's OUTPUT features, output slots = its INPUT slots.  Layers whose inputs include the encoding (4: the skip
    # connection, 0) send those 48 slots through a 3-output-tile "tail" ([T = 3][lane][r] k tiles, 8 per chunk) - in the kernel's
    # order: 7, 6, 5, tail of 4, 4, 3, 2, 1, tail of 0.
    def transposed(l, kfeat, ofeat):
        return flat.mat_index(f"w{l}", kfeat, ofeat).T                         # [out slot, k slot]

    def tail_chunks(midx):                                                      # midx [48, 256]
        kts = [_ktile_index_n(midx, t, 3) for t in range(midx.shape[1] // 16)]
        return [np.concatenate(kts[c: c + 8]) for c in range(0, len(kts), 8)]

    rev = []
    hw = W - enc
    for l in range(D - 1, 0, -1):
        out_dim, in_dim = dims[l]
        kfeat = _pad(ar(out_dim), (out_dim + 15) // 16 * 16)
        if l in skips:
            rev += tail_chunks(transposed(l, kfeat, np.where(ENC_SLOTS >= 0, hw + ENC_SLOTS, -1)))
            ofeat = _pad(ar(hw), 256)
        else:
            ofeat = _pad(ar(in_dim), 256)
        rev += _layer_chunks(transposed(l, kfeat, ofeat), len(kfeat) // 16, 0)
    rev += tail_chunks(transposed(0, ar(W), ENC_SLOTS))
    return PackPlan(PROG_SURFACE, flat, chunks, aux, chunks_rev=rev)


def radiance_plan(view_tiles: int, W: int = 256, D: int = 4, W_geo_feat: int = 256) -> PackPlan:
    """Geometry-feature rows of the SDF net's last layer + RadianceNet (models/base.py:312-391)
    -> program for k_radiance<view_tiles>.  view_tiles = 1: raw view dirs (VolSDF configs, input
    [x, v, n, feat] = 265); 3: embed_multires_view = 4 (NeuS configs, 289)."""
    if not (W == 256 and D == 4 and W_geo_feat == 256 and view_tiles in (1, 3)):
        raise NotImplementedError("gfx950 radiance kernel is built for W=256, D=4, W_geo_feat=256, "
                                  "embed_multires=-1, embed_multires_view in (-1, 4)")
    n_extra = 9 if view_tiles == 1 else 33
    flat = _Flat()
    flat.add("w8", (1 + W_geo_feat, 256)); flat.add("b8", (1 + W_geo_feat,))
    in0 = n_extra + W_geo_feat
    rdims = [(W, in0)] + [(W, W)] * (D - 1) + [(3, W)]
    for l, (o, i) in enumerate(rdims):
        flat.add(f"r{l}", (o, i)); flat.add(f"rb{l}", (o,))
    ar = np.arange
    chunks = []
    chunks += _layer_chunks(flat.mat_index("w8", 1 + ar(256), ar(256)), 16, 0)
    cols0 = np.concatenate([n_extra + ar(256), _pad(ar(n_extra), 16 * view_tiles)])
    chunks += _layer_chunks(flat.mat_index("r0", ar(256), cols0), 16, view_tiles)
    for l in range(1, D):
        chunks += _layer_chunks(flat.mat_index(f"r{l}", ar(256), ar(256)), 16, 0)
    aux = [flat.vec_index("b8", 1 + ar(256))]
    for l in range(D):
        aux.append(flat.vec_index(f"rb{l}", ar(256)))
    aux.append(flat.mat_index(f"r{D}", ar(3), ar(256)).reshape(-1))
    aux.append(flat.vec_index(f"rb{D}", _pad(ar(3), 4)))
    aux = np.concatenate(aux)
    assert len(aux) == RAD_AUX_FLOATS
    return PackPlan(PROG_RADIANCE, flat, chunks, aux)


# ----------------------------------------------------------------------------------------------------
# split-bf16 ("bf16x3") programs for csrc/mlp_chain_bf16.hip
# ----------------------------------------------------------------------------------------------------
PROG_SURFACE_BF16 = 3
PROG_RADIANCE_BF16 = 4
TS_FLOATS = 512           # one (k-step, output tile) of a chunk: (hi, lo) x 64 lanes x 8 bf16 = 2 KiB
TERM_WORD = {"bf16": 1, "fp16": 2}      # header word 10 (ABI 3): the fragment encoding of a split blob - bf16 hi + lo / fp16 hi + lo


def unit_feature_hidden(ks: int, g: int, e: int) -> int:
    """Slot (k-step = unit ks, lane group g, element e) of a hidden-layer input -> feature index.  The C layout of
    v_mfma_f32_16x16x32_bf16 puts row 4g + r of output tile T in register r; unit U of the next layer's input
    is (tile 2U regs 0..3, tile 2U+1 regs 0..3)."""
    return 32 * ks + (4 * g + e if e < 4 else 16 + 4 * g + (e - 4))


def unit_feature_enc(q: int, g: int, e: int) -> int:
    """Slot of the 2 positional-encoding units -> reference feature (0..38) or -1.  Lane group g < 3 owns
    coordinate g: local index m = 8q + e: 0 raw, 1 + 2k sin(2^k x_g), 2 + 2k cos(2^k x_g) (k < 6); group 3 pads."""
    m = 8 * q + e
    if g >= 3 or m > 12:
        return -1
    if m == 0:
        return g
    k, is_cos = divmod(m - 1, 2)
    return 3 + 6 * k + 3 * is_cos + g


def unit_feature_extra(q: int, g: int, e: int, n_extra: int) -> int:
    m = 32 * q + 8 * g + e
    return m if m < n_extra else -1


def _kstep_index(flat, name, out_dim, ks, cols_fn, row0=0):
    """Index array [T=16][lane=64][e=8] of one k-step: W[row0 + 16T + i][cols_fn(ks, g, e)], lane = 16g + i
    (rows >= out_dim: zero)."""
    R, C = flat.shape[name]
    idx = np.full((16, 64, 8), flat.zero, dtype=np.int64)
    i = np.arange(16)
    for T in range(16):
        rows = 16 * T + i
        ok = rows < out_dim
        for g in range(4):
            for e in range(8):
                c = cols_fn(ks, g, e)
                if c < 0:
                    continue
                idx[T, (16 * g + i)[ok], e] = flat.base[name] + (row0 + rows[ok]) * C + c
    return idx.reshape(-1)


CHUNK_KS = 2              # k-steps per chunk (64 KiB)


def _layer_chunks_bf16(flat, name, out_dim, cols_fn, nu_base, nu_extra, row0=0, chunk_ks=CHUNK_KS):
    """ceil(nu_base/2) chunks of base k-steps, then one chunk with the (<= 2) extra k-steps (matches run_layer in
    mlp_chain_bf16.hip)."""
    assert nu_extra <= chunk_ks
    chunks = []
    for c0 in range(0, nu_base, chunk_ks):
        chunks.append(np.concatenate([_kstep_index(flat, name, out_dim, ks, cols_fn, row0) for ks in range(c0, min(c0 + chunk_ks, nu_base))]))
    if nu_extra:
        chunks.append(np.concatenate([_kstep_index(flat, name, out_dim, nu_base + x, cols_fn, row0) for x in range(nu_extra)]))
    return chunks


class PackPlanBF16(PackPlan):
    """Chunks hold bf16 hi/lo fragments, k-outer: float[k-step][T=16][term=2][lane=64][4] (= 8 bf16 per lane).

    chunk_mul: optional per-chunk second gather (same length as the chunk's index) whose values multiply the
    first (used to fold a row vector into a transposed matrix); chunk_scale: optional per-chunk constant.
    nc_main: number of leading chunks that form the forward program (header word 2); header word 6 = all chunks."""

    def __init__(self, prog, flat, chunk_indices, aux, chunk_mul=None, chunk_scale=None, nc_main=None, nc_cycle=None, table2=None):
        self.prog, self.flat = prog, flat
        offs = [HDR_INTS]
        for c in chunk_indices:
            offs.append(offs[-1] + (len(c) // 512) * TS_FLOATS)
        self.nc_all = len(chunk_indices) if nc_cycle is None else nc_cycle      # header word 6: chunks of the second program
        self.nc = self.nc_all if nc_main is None else nc_main
        assert len(chunk_indices) + 1 <= 112
        self.aux_off = offs[-1]
        self.cindex = np.concatenate(chunk_indices)
        self.cmul = None
        if chunk_mul is not None:
            one = flat.zero + 1                                  # pack() appends [0.0, 1.0] to the source vector
            self.cmul = np.concatenate([m if m is not None else np.full(len(c), one, dtype=np.int64)
                                        for m, c in zip(chunk_mul, chunk_indices)])
        self.cscale = None
        if chunk_scale is not None:
            self.cscale = np.concatenate([np.full(len(c), sc, dtype=np.float32) for sc, c in zip(chunk_scale, chunk_indices)])
        self.aindex = aux
        # the kernels' weight stream always copies 64 KiB per chunk: zero padding keeps the copy of a short last
        # chunk inside the blob
        self.total = max(self.aux_off + len(aux), offs[-2] + CHUNK_KS * 16 * TS_FLOATS)
        self.pad = self.total - (self.aux_off + len(aux))
        hdr = np.zeros(HDR_INTS, dtype=np.int32)
        hdr[0], hdr[1], hdr[2], hdr[3], hdr[4], hdr[5], hdr[6] = MAGIC, prog, self.nc, self.total, self.aux_off, len(aux), self.nc_all
        hdr[HDR_OFFS: HDR_OFFS + len(offs)] = offs
        # optional third program: an explicit list of chunk indices (header word 7 = its length, offsets at words 128..)
        if table2 is not None:
            assert len(table2) <= 120
            hdr[7] = len(table2)
            hdr[128: 128 + len(table2)] = [offs[c] for c in table2]
        self.header = hdr
        self._index_t = {}

    def pack(self, tensors: dict) -> torch.Tensor:
        dev = tensors[self.flat.names[0]].device
        for n in self.flat.names:
            assert tuple(tensors[n].shape) == self.flat.shape[n], n
        scale = getattr(self, "scale", {})
        parts = [(tensors[n].detach().to(torch.float32) * scale[n] if n in scale else tensors[n].detach().to(torch.float32)).reshape(-1)
                 for n in self.flat.names]
        parts.append(torch.tensor([0.0, 1.0], dtype=torch.float32, device=dev))
        src = torch.cat(parts)
        key = str(dev)
        if key not in self._index_t:
            self._index_t[key] = (torch.from_numpy(self.cindex).to(dev), torch.from_numpy(self.aindex).to(dev),
                                  None if self.cmul is None else torch.from_numpy(self.cmul).to(dev),
                                  None if self.cscale is None else torch.from_numpy(self.cscale).to(dev))
        ci, ai, cm, cs = self._index_t[key]
        w = src[ci]
        if cm is not None:
            w = w * src[cm]
        if cs is not None:
            w = w * cs
        w = w.reshape(-1, 64, 8)                              # [(k-step, tile) of all chunks, lane, e]
        dt = torch.float16 if getattr(self, "term", "bf16") == "fp16" else torch.bfloat16      # "fp16": the 2-MFMA variant (precision 4)
        hi = w.to(dt)                                         # round-to-nearest-even
        lo = (w - hi.to(torch.float32)).to(dt)
        body = torch.stack([hi, lo], dim=1).contiguous().view(torch.float32).reshape(-1)   # [..][term][lane][4]
        hdr = torch.from_numpy(self.header.copy()).view(torch.float32).to(dev)
        return torch.cat([hdr, body, src[ai], torch.zeros(self.pad, dtype=torch.float32, device=dev)]).contiguous()


def _kstep_index_T(flat, name, ks, kfeat_fn, row_feature, tiles=range(16), row0=0):
    """Transposed gather for the reverse-mode chain: index array [T in tiles][lane=64][e=8] of one k-step with
    A[row][k] = W[kfeat][rowfeat]: W's ROW index comes from the k slot (kfeat_fn(ks, g, e), -1 = zero), W's COLUMN
    index from the output row 16T + i (row_feature[16T + i], -1 = zero)."""
    R, C = flat.shape[name]
    tiles = list(tiles)
    idx = np.full((len(tiles), 64, 8), flat.zero, dtype=np.int64)
    i = np.arange(16)
    for n, T in enumerate(tiles):
        rf = row_feature[16 * T + i]
        ok = rf >= 0
        for g in range(4):
            for e in range(8):
                kf = kfeat_fn(ks, g, e)
                if kf < 0 or kf + row0 >= R:
                    continue
                idx[n, (16 * g + i)[ok], e] = flat.base[name] + (kf + row0) * C + rf[ok]
    return idx.reshape(-1)


def _kstep_mul_index(flat, name, row, ks, kfeat_fn, ntiles=16):
    """Second gather matching _kstep_index_T: W2[row][kfeat] for every element (the value depends on the k slot only)."""
    R, C = flat.shape[name]
    idx = np.full((ntiles, 64, 8), flat.zero + 1, dtype=np.int64)
    for g in range(4):
        for e in range(8):
            kf = kfeat_fn(ks, g, e)
            if 0 <= kf < C:
                idx[:, 16 * g: 16 * g + 16, e] = flat.base[name] + row * C + kf
    return idx.reshape(-1)


# ---- one-wave-per-SIMD K2 (csrc/mlp_k2_w32.hip): v_mfma_f32_32x32x16_bf16 layouts ---------------------------------------------
# A: lane l holds 8 k-slots of row l & 31 (half h = l >> 5: slots of k-step sub-block h); B: lane (col = l & 31, h) holds 8
# k-slots of its column; C: reg r of tile T <-> row 32 T + 8 (r >> 2) + 4 h + (r & 3).  The 16 registers of output tile T become
# units (16-deep k-steps) 2 T (regs 0..7) and 2 T + 1 (regs 8..15) of the next layer.
W32_CHUNK_KS = 4          # 16-deep k-steps per 64 KiB chunk: 4 x 8 output tiles x (hi, lo) x 1 KiB


def w32_feature_hidden(ks: int, h: int, e: int) -> int:
    return 16 * ks + 8 * (e >> 2) + 4 * h + (e & 3)


def w32_feature_enc(q: int, h: int, e: int) -> int:
    """Slot (unit q < 3, half h, element e) of the 3 positional-encoding units -> reference feature (0..38) or -1.  Half 0 holds
    x, y, z and the (sin, cos) pairs of bands k = 0..2, half 1 the pairs of bands k = 3..5: a lane evaluates 9 sincos."""
    m = 8 * q + e
    if h == 0:
        if m < 3:
            return m
        m -= 3
    if m >= 18:
        return -1
    kc, is_cos = divmod(m, 2)
    k, c = divmod(kc, 3)
    return 3 + 6 * (k + 3 * h) + 3 * is_cos + c


def _item_index_w32(flat, name, out_dim, T, ks, cols_fn):
    """Index array [lane=64][e=8] of item (output tile T of 32 rows, k-step ks): W[32 T + (lane & 31)][cols_fn(ks, lane >> 5, e)]."""
    R, C = flat.shape[name]
    idx = np.full((64, 8), flat.zero, dtype=np.int64)
    lane = np.arange(64)
    rows = 32 * T + (lane & 31)
    for h in range(2):
        sel = (rows < out_dim) & ((lane >> 5) == h)
        for e in range(8):
            c = cols_fn(ks, h, e)
            if c >= 0:
                idx[sel, e] = flat.base[name] + rows[sel] * C + c
    return idx.reshape(-1)


def _layer_chunks_w32(flat, name, out_dim, cols_fn, nks):
    """k-step-major chunks of W32_CHUNK_KS k-steps; inside a chunk item = k-step-in-chunk * 8 + output tile."""
    chunks = []
    for c0 in range(0, nks, W32_CHUNK_KS):
        chunks.append(np.concatenate([_item_index_w32(flat, name, out_dim, T, ks, cols_fn)
                                      for ks in range(c0, min(c0 + W32_CHUNK_KS, nks)) for T in range(8)]))
    return chunks


D_UNORM = 65535.0         # softplus'(z) in [0, 1] is handed from the forward to the backward sweep as unorm16
GRAD_ENC_ROW0 = 217       # rows 217..255 of the backward outputs of layers 4 and 0 carry d sdf / d enc[0..38]


def surface_plan_bf16(W: int = 256, D: int = 8, skips=(4,), multires: int = 6, W_geo_feat: int = 256, term: str = "bf16") -> PackPlanBF16:
    """term = "fp16": the same programs with fp16 hi + lo weight fragments for the 2-MFMA kernels (csrc/mlp_chain_f16x2.hip, C-ABI
    precision 4); the reverse-mode chunks then do NOT absorb the unorm16 scale 1 / 65535 (w / 65535 would be an fp16 subnormal:
    the kernel applies it to softplus' instead)."""
    if not (W == 256 and D == 8 and tuple(skips) == (4,) and multires == 6 and W_geo_feat == 256):
        raise NotImplementedError("gfx950 surface kernels are built for W=256, D=8, skips=[4], embed_multires=6, W_geo_feat=256")
    enc = 39
    flat = _Flat()
    dims = []
    for l in range(D + 1):
        out_dim = (1 + W_geo_feat) if l == D else (W - enc if (l + 1) in skips else W)
        in_dim = enc if l == 0 else W
        dims.append((out_dim, in_dim))
        flat.add(f"w{l}", (out_dim, in_dim)); flat.add(f"b{l}", (out_dim,))
    hw = W - enc
    chunks = []
    for l in range(D):
        out_dim, in_dim = dims[l]
        if l == 0:
            chunks += _layer_chunks_bf16(flat, f"w{l}", out_dim, unit_feature_enc, 2, 0)
        elif l in skips:
            # input = 7 hidden units (217 features), then the 2 encoding units (the first one closes the base
            # chunks, the second is the extra chunk)
            def fn(ks, g, e, hw=hw):
                if ks < 7:
                    f = unit_feature_hidden(ks, g, e)
                    return f if f < hw else -1
                f = unit_feature_enc(ks - 7, g, e)
                return hw + f if f >= 0 else -1
            chunks += _layer_chunks_bf16(flat, f"w{l}", out_dim, fn, 8, 1)
        else:
            def fn(ks, g, e, in_dim=in_dim):
                f = unit_feature_hidden(ks, g, e)
                return f if f < in_dim else -1
            chunks += _layer_chunks_bf16(flat, f"w{l}", out_dim, fn, 8, 0)
    ar = np.arange
    nc_fwd = len(chunks)
    # ---- reverse-mode program (k_sdf_grad_bf16): d sdf / d a_{l-1} = W_l^T (d sdf / d a_l * softplus'(z_l)), l = 7..0.
    # Transposed matrices, k slots in the same unit order (the backward inputs are the previous backward step's
    # accumulators), output rows = the layer's input features in natural order.  Layer 7 absorbs the sdf row of
    # the last linear layer (its input units are softplus'(z_7) alone); layers 6..0 absorb 1/65535 (unorm16
    # softplus' from the forward sweep); layer 4's 1/sqrt(2) comes with plan.scale.  Layer 0 only has the 39
    # encoding rows: they sit at rows 217..255 like layer 4's (3 output tiles, all 8 k-steps in one 48 KiB chunk).
    mul, scl = [None] * nc_fwd, [1.0] * nc_fwd
    nat = ar(256)
    for l in range(D - 1, 0, -1):
        out_dim, in_dim = dims[l]

        def kf(ks, g, e, out_dim=out_dim):
            f = unit_feature_hidden(ks, g, e)
            return f if f < out_dim else -1
        for c0 in range(0, 8, CHUNK_KS):
            chunks.append(np.concatenate([_kstep_index_T(flat, f"w{l}", ks, kf, nat) for ks in range(c0, c0 + CHUNK_KS)]))
            mul.append(np.concatenate([_kstep_mul_index(flat, f"w{D}", 0, ks, kf) for ks in range(c0, c0 + CHUNK_KS)]) if l == D - 1 else None)
            scl.append(1.0 if (l == D - 1 or term == "fp16") else 1.0 / D_UNORM)
    enc_rows = np.where(nat >= GRAD_ENC_ROW0, nat - GRAD_ENC_ROW0, -1)
    chunks.append(np.concatenate([_kstep_index_T(flat, "w0", ks, unit_feature_hidden, enc_rows, tiles=(13, 14, 15)) for ks in range(8)]))
    mul.append(None)
    scl.append(1.0 if term == "fp16" else 1.0 / D_UNORM)
    aux = [flat.vec_index(f"b{l}", _pad(ar(dims[l][0]), 256)) for l in range(D)]
    aux.append(flat.mat_index(f"w{D}", np.array([0]), ar(256)).reshape(-1))
    aux.append(flat.vec_index(f"b{D}", _pad(np.array([0]), 4)))
    aux = np.concatenate(aux)
    assert len(aux) == SURF_AUX_FLOATS
    # ---- second-order program (k_sdf_bwd2_bf16, the fine-tune step's SDF backward): the same transposed layers, but
    # layer 7 WITHOUT the folded sdf row (its cotangent is an input) and with the 1/65535 of the others
    nc_grad = len(chunks)

    def kf7(ks, g, e):
        return unit_feature_hidden(ks, g, e)
    for c0 in range(0, 8, CHUNK_KS):
        chunks.append(np.concatenate([_kstep_index_T(flat, f"w{D - 1}", ks, kf7, nat) for ks in range(c0, c0 + CHUNK_KS)]))
        mul.append(None)
        scl.append(1.0 / D_UNORM)
    table2 = list(range(nc_grad, nc_grad + 4)) + list(range(nc_fwd + 4, nc_fwd + 4 + 24))      # B7', then B6..B1
    # ---- fourth program: the forward layers again in the 32x32x16 fragment layout of the one-wave-per-SIMD K2 (mlp_k2_w32.hip)
    w32_first = len(chunks)
    for l in range(D):
        out_dim, in_dim = dims[l]
        if l == 0:
            w32 = _layer_chunks_w32(flat, f"w{l}", out_dim, w32_feature_enc, 3)
        elif l in skips:
            def fn(ks, h, e, hw=hw):
                if ks < 14:
                    f = w32_feature_hidden(ks, h, e)
                    return f if f < hw else -1
                f = w32_feature_enc(ks - 14, h, e)
                return hw + f if f >= 0 else -1
            w32 = _layer_chunks_w32(flat, f"w{l}", out_dim, fn, 17)
        else:
            def fn(ks, h, e, in_dim=in_dim):
                f = w32_feature_hidden(ks, h, e)
                return f if f < in_dim else -1
            w32 = _layer_chunks_w32(flat, f"w{l}", out_dim, fn, 16)
        chunks += w32
        mul += [None] * len(w32)
        scl += [1.0] * len(w32)
    plan = PackPlanBF16(PROG_SURFACE_BF16, flat, chunks, aux, chunk_mul=mul, chunk_scale=scl, nc_main=nc_fwd, nc_cycle=nc_grad,
                        table2=table2)
    plan.header[8], plan.header[9] = len(chunks) - w32_first, w32_first      # the w32 program: chunk count, first chunk
    # cat[h, enc] / sqrt(2) (base.py:250) is applied to the skip layer's weights instead of its inputs
    plan.scale = {f"w{l}": 1.0 / float(np.sqrt(2.0)) for l in skips}
    plan.term = term
    plan.header[10] = TERM_WORD[term]
    return plan


def radiance_plan_bf16(view_tiles: int, W: int = 256, D: int = 4, W_geo_feat: int = 256, term: str = "bf16") -> PackPlanBF16:
    if not (W == 256 and D == 4 and W_geo_feat == 256 and view_tiles in (1, 3)):
        raise NotImplementedError("gfx950 radiance kernel is built for W=256, D=4, W_geo_feat=256")
    n_extra = 9 if view_tiles == 1 else 33
    flat = _Flat()
    flat.add("w8", (1 + W_geo_feat, 256)); flat.add("b8", (1 + W_geo_feat,))
    in0 = n_extra + W_geo_feat
    rdims = [(W, in0)] + [(W, W)] * (D - 1) + [(3, W)]
    for l, (o, i) in enumerate(rdims):
        flat.add(f"r{l}", (o, i)); flat.add(f"rb{l}", (o,))
    # layer A: feat = W8[1:257] h7  (rows 1.. of w8)
    chunks = _layer_chunks_bf16(flat, "w8", 256, unit_feature_hidden, 8, 0, row0=1)
    extra_units = 1 if view_tiles == 1 else 2

    def fn0(ks, g, e):
        if ks < 8:
            return n_extra + unit_feature_hidden(ks, g, e)
        return unit_feature_extra(ks - 8, g, e, n_extra)
    chunks += _layer_chunks_bf16(flat, "r0", W, fn0, 8, extra_units)
    for l in range(1, D):
        chunks += _layer_chunks_bf16(flat, f"r{l}", W, unit_feature_hidden, 8, 0)
    ar = np.arange
    aux = [flat.vec_index("b8", 1 + ar(256))]
    for l in range(D):
        aux.append(flat.vec_index(f"rb{l}", ar(256)))
    aux.append(flat.mat_index(f"r{D}", ar(3), ar(256)).reshape(-1))
    aux.append(flat.vec_index(f"rb{D}", _pad(ar(3), 4)))
    aux = np.concatenate(aux)
    assert len(aux) == RAD_AUX_FLOATS
    # ---- reverse program (k_radiance_bwd_bf16): g_r{l-1} = R_l^T (g_r{l} * [r_l > 0]) for l = 3..1, the normal rows of
    # R0^T (one output tile, all 8 k-steps in one 16 KiB chunk), the feature rows of R0^T, then W8[1:]^T.
    nc_fwd = len(chunks)
    nat = ar(256)
    for name in ("r3", "r2", "r1"):
        for c0 in range(0, 8, CHUNK_KS):
            chunks.append(np.concatenate([_kstep_index_T(flat, name, ks, unit_feature_hidden, nat) for ks in range(c0, c0 + CHUNK_KS)]))
    nrm_rows = np.full(256, -1, dtype=np.int64)
    nrm_rows[:3] = n_extra - 3 + ar(3)
    chunks.append(np.concatenate([_kstep_index_T(flat, "r0", ks, unit_feature_hidden, nrm_rows, tiles=(0,)) for ks in range(8)]))
    for c0 in range(0, 8, CHUNK_KS):
        chunks.append(np.concatenate([_kstep_index_T(flat, "r0", ks, unit_feature_hidden, n_extra + nat) for ks in range(c0, c0 + CHUNK_KS)]))
    for c0 in range(0, 8, CHUNK_KS):
        chunks.append(np.concatenate([_kstep_index_T(flat, "w8", ks, unit_feature_hidden, nat, row0=1) for ks in range(c0, c0 + CHUNK_KS)]))
    plan = PackPlanBF16(PROG_RADIANCE_BF16, flat, chunks, aux, nc_main=nc_fwd)
    plan.term = term
    plan.header[10] = TERM_WORD[term]
    return plan


def fold_weight_norm(weight_g: torch.Tensor, weight_v: torch.Tensor) -> torch.Tensor:
    """w[o,:] = g[o] * v[o,:] / ||v[o,:]||  (torch.nn.utils.weight_norm default dim=0; the
    reference re-does this on every forward, base.py:226-227 - here once per weight update)."""
    return torch._weight_norm(weight_v, weight_g, 0)


def surface_tensors(sd: dict, stem: str = "implicit_surface.surface_fc_layers", D: int = 8) -> dict:
    out = {}
    for l in range(D + 1):
        out[f"w{l}"] = fold_weight_norm(sd[f"{stem}.{l}.weight_g"], sd[f"{stem}.{l}.weight_v"])
        out[f"b{l}"] = sd[f"{stem}.{l}.bias"]
    return out


def radiance_tensors(sd: dict, surf_stem: str = "implicit_surface.surface_fc_layers",
                     rad_stem: str = "radiance_net.layers", D_surf: int = 8, D: int = 4) -> dict:
    out = {
        "w8": fold_weight_norm(sd[f"{surf_stem}.{D_surf}.weight_g"], sd[f"{surf_stem}.{D_surf}.weight_v"]),
        "b8": sd[f"{surf_stem}.{D_surf}.bias"],
    }
    for l in range(D + 1):
        out[f"r{l}"] = fold_weight_norm(sd[f"{rad_stem}.{l}.weight_g"], sd[f"{rad_stem}.{l}.weight_v"])
        out[f"rb{l}"] = sd[f"{rad_stem}.{l}.bias"]
    return out
