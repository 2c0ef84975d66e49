"""`batchify_query` (row a10; the call shape of the reference's utils/train_util.py:23-75): evaluate a point query on
[(B), N_rays, N_pts, ...] tensors in chunks of the flattened ray x point axis.

The HIP renderer never materialises its points, so nothing inside this package needs this; it is
exported for the reference's own consumers of the point-query boundary B2 - `volume_render`-style code, the surface
renderer (models/ray_casting.py) and mesh extraction - which call `batchify_query(model.forward, pts, view_dirs,
chunk=netchunk, dim_batchify=..., return_nablas=...)`.  With the HIP-backed `model.forward` one chunk can hold tens of
millions of points; `chunk` is honoured as given.
"""
import torch


def _merge(parts, dim, n_rays, n_pts):
    """Chunk results of one output slot -> [(B), N_rays, N_pts, ...]; dict outputs are merged key by key."""
    if isinstance(parts[0], dict):
        return {k: _merge([p[k] for p in parts], dim, n_rays, n_pts) for k in parts[0]}
    v = parts[0] if len(parts) == 1 else torch.cat(parts, dim=dim)
    return v.reshape(*v.shape[:dim], n_rays, n_pts, *v.shape[dim + 1:])


def batchify_query(query_fn, *args, chunk, dim_batchify, return_nablas):
    """query_fn(*flat_args_chunk, return_nablas=...) on chunks of `chunk` points along the flattened (dim_batchify, dim_batchify+1)
    axes of every non-None tensor in `args`.  Returns the single output, or a tuple of outputs each reshaped back to
    [..., N_rays, N_pts, ...]; a two-output query gets a trailing None (the reference's placeholder for the nablas a
    (radiance, sdf) query does not return)."""
    if dim_batchify not in (0, 1, 2):
        raise NotImplementedError(f"dim_batchify = {dim_batchify}")
    n_rays, n_pts = args[0].shape[dim_batchify], args[0].shape[dim_batchify + 1]
    flat = [a.flatten(dim_batchify, dim_batchify + 1) for a in args if a is not None]
    per_chunk = []
    for pieces in zip(*(a.split(chunk, dim=dim_batchify) for a in flat)):
        r = query_fn(*pieces, return_nablas=return_nablas)
        per_chunk.append(r if isinstance(r, tuple) else (r,))
    outs = [_merge(list(slot), dim_batchify, n_rays, n_pts) for slot in zip(*per_chunk)]
    if len(outs) == 1:
        return outs[0]
    if len(outs) == 2:
        outs.append(None)
    return tuple(outs)
