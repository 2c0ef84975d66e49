"""batchify_query (row a10): the call shape of utils/train_util.py:23-75 - against direct evaluation, and (build container
only) against the reference's own function."""
import os
import sys

import pytest
import torch


def _q3(x, v, return_nablas=False):
    rad = x + v                                    # exact elementwise ops: chunked == direct, bit for bit
    sdf = x[..., 0] * 2.0 - 1.0
    if return_nablas:
        return rad, sdf, x * 0.5
    return rad, sdf


def _qdict(x, return_nablas=False):
    return {"a": x * 2.0, "b": x.sum(-1)}, x[..., 0]


@pytest.mark.parametrize("dim", [0, 1, 2])
def test_batchify_query_equals_direct_evaluation(dim):
    from nerfart_amd.train_util import batchify_query
    g = torch.Generator().manual_seed(dim)
    lead = (2, 3)[:dim]
    x, v = torch.randn(*lead, 7, 5, 3, generator=g), torch.randn(*lead, 7, 5, 3, generator=g)
    rad, sdf, nab = batchify_query(_q3, x, v, chunk=8, dim_batchify=dim, return_nablas=True)
    r0, s0, n0 = _q3(x, v, True)
    assert rad.shape == (*lead, 7, 5, 3) and sdf.shape == (*lead, 7, 5)
    assert torch.equal(rad, r0) and torch.equal(sdf, s0) and torch.equal(nab, n0)
    out = batchify_query(_q3, x, v, chunk=1000, dim_batchify=dim, return_nablas=False)           # two outputs: None appended
    assert len(out) == 3 and out[2] is None and torch.equal(out[1], s0)
    d, first = batchify_query(_qdict, x, None, chunk=4, dim_batchify=dim, return_nablas=False)[:2]  # None args dropped, dict outputs
    assert torch.equal(d["a"], x * 2.0) and torch.equal(d["b"], x.sum(-1)) and torch.equal(first, x[..., 0])
    single = batchify_query(lambda p, return_nablas: p.sum(-1), x, chunk=6, dim_batchify=dim, return_nablas=False)
    assert torch.equal(single, x.sum(-1))
    with pytest.raises(NotImplementedError):
        batchify_query(_q3, x, v, chunk=8, dim_batchify=3, return_nablas=False)


@pytest.mark.skipif(not os.path.exists("/root/reference/utils/train_util.py"), reason="reference tree not present (GPU box)")
def test_batchify_query_matches_the_reference_function():
    import importlib.util
    import types
    stubs = {}
    for name in ("utils", "utils.print_fn", "utils.logger"):
        if name not in sys.modules:
            stubs[name] = sys.modules[name] = types.ModuleType(name)
    if "utils.print_fn" in stubs:
        stubs["utils.print_fn"].log = types.SimpleNamespace(info=print)
    try:
        spec = importlib.util.spec_from_file_location("_ref_train_util", "/root/reference/utils/train_util.py")
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
    finally:
        for name in stubs:
            sys.modules.pop(name, None)
    from nerfart_amd.train_util import batchify_query
    g = torch.Generator().manual_seed(5)
    x, v = torch.randn(1, 9, 4, 3, generator=g), torch.randn(1, 9, 4, 3, generator=g)
    for rn in (True, False):
        a = batchify_query(_q3, x, v, chunk=7, dim_batchify=1, return_nablas=rn)
        b = ref.batchify_query(_q3, x, v, chunk=7, dim_batchify=1, return_nablas=rn)
        assert len(a) == len(b) == 3
        for p, q in zip(a, b):
            assert (p is None and q is None) or torch.equal(p, q)
