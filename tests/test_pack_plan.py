"""The C library's weight-blob layout (csrc/pack_blob.hip: `chunk_desc` / `elem_src`, closed forms evaluated per element by the pack kernels)
against nerfart_amd/packing.py's numpy plans (the source of truth of the CPU emulation, tests/emul_chain.py, which walks the real blob and
must reproduce the oracle): `nerfart_pack_plan_debug` evaluates the library's functions on the HOST into plain tables, and they must equal
packing.py's entry for entry - every gather index of every chunk, the second-factor gather, the per-chunk scale, the aux gather, and all
512 header words.  No GPU needed: this is integer work.  (GPU: tests/test_gpu_pack.py holds the packed blobs' VALUES to packing.py's.)"""
import ctypes as C
import os

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    l = C.CDLL(os.path.join(REPO, "nerfart_amd", "csrc", "libnerfart_hip.so"))
    l.nerfart_pack_plan_debug.restype = C.c_int
    l.nerfart_pack_plan_debug.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6
    for f in ("nerfart_surface_blob_floats", "nerfart_radiance_blob_floats"):
        getattr(l, f).restype = C.c_longlong
        getattr(l, f).argtypes = [C.c_int, C.c_int]
    return l


def _debug(lib, prog, view_tiles, fp16):
    sizes = (C.c_longlong * 4)()
    assert lib.nerfart_pack_plan_debug(prog, view_tiles, fp16, sizes, None, None, None, None, None) == 0
    n_c, n_a, total, n_chunks = (int(x) for x in sizes)
    hdr = np.zeros(512, np.int32); ci = np.zeros(n_c, np.int32); cm = np.zeros(n_c, np.int32); cs = np.zeros(n_c, np.float32); ai = np.zeros(n_a, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.nerfart_pack_plan_debug(prog, view_tiles, fp16, sizes, p(hdr), p(ci), p(cm), p(cs), p(ai)) == 0
    return dict(header=hdr, cindex=ci, cmul=cm, cscale=cs, aindex=ai, total=total, n_chunks=n_chunks)


@pytest.mark.parametrize("view_tiles", [1, 3])
def test_fp32_programs_equal_the_numpy_plans(lib, view_tiles):
    from nerfart_amd import packing
    for prog, plan in ((1, packing.surface_plan()), (2, packing.radiance_plan(view_tiles))):
        d = _debug(lib, prog, view_tiles, 0)
        n_aux = int(plan.header[5])
        np.testing.assert_array_equal(d["header"], plan.header, err_msg=f"program {prog}: header")
        np.testing.assert_array_equal(d["cindex"], plan.index[:-n_aux], err_msg=f"program {prog}: chunk gather")
        np.testing.assert_array_equal(d["aindex"], plan.index[-n_aux:], err_msg=f"program {prog}: aux gather")
        assert d["total"] == plan.total and d["n_chunks"] == plan.nc_all
        assert (d["cscale"] == 1.0).all() and (d["cmul"] == plan.flat.zero + 1).all()


@pytest.mark.parametrize("term", ["bf16", "fp16"])
@pytest.mark.parametrize("view_tiles", [1, 3])
def test_split_programs_equal_the_numpy_plans(lib, view_tiles, term):
    from nerfart_amd import packing
    for prog, plan in ((3, packing.surface_plan_bf16(term=term)), (4, packing.radiance_plan_bf16(view_tiles, term=term))):
        d = _debug(lib, prog, view_tiles, int(term == "fp16"))
        np.testing.assert_array_equal(d["header"], plan.header, err_msg=f"program {prog}: header")
        np.testing.assert_array_equal(d["cindex"], plan.cindex, err_msg=f"program {prog}: chunk gather")
        np.testing.assert_array_equal(d["aindex"], plan.aindex, err_msg=f"program {prog}: aux gather")
        one = plan.flat.zero + 1
        np.testing.assert_array_equal(d["cmul"], plan.cmul if plan.cmul is not None else np.full(len(plan.cindex), one), err_msg=f"program {prog}: second factor")
        np.testing.assert_array_equal(d["cscale"], plan.cscale if plan.cscale is not None else np.ones(len(plan.cindex), np.float32), err_msg=f"program {prog}: chunk scale")
        assert d["total"] == plan.total
        assert int(d["header"][10]) == packing.TERM_WORD[term]


def test_blob_sizes_by_precision(lib):
    from nerfart_amd import packing
    assert lib.nerfart_surface_blob_floats(0, 6) == packing.surface_plan().total
    assert lib.nerfart_surface_blob_floats(1, 6) == lib.nerfart_surface_blob_floats(4, 6) == packing.surface_plan_bf16().total
    for vt in (1, 3):
        assert lib.nerfart_radiance_blob_floats(0, vt) == packing.radiance_plan(vt).total
        assert lib.nerfart_radiance_blob_floats(1, vt) == packing.radiance_plan_bf16(vt).total
    assert lib.nerfart_surface_blob_floats(2, 6) == 0 and lib.nerfart_surface_blob_floats(1, 4) == 0 and lib.nerfart_radiance_blob_floats(1, 2) == 0
