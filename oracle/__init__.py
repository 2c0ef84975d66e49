"""CPU oracle for the NeRF-Art hot path (VolSDF / NeuS volumetric renderer).

TEST INFRASTRUCTURE ONLY.  This package is a plain PyTorch-CPU fp32
restatement of the reference algorithm (cassiePython/NeRF-Art), written
from the reference's behaviour, each function citing the reference
file:line it follows.  It exists to *check* the HIP path, never to be it:

  * only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
    leg of ``bench.py`` may import it;
  * nothing under ``nerfart_amd/`` imports it (tests/test_boundary.py
    greps for that);
  * it is pinned against golden vectors captured from the real reference
    running in the build container (``tests/golden/make_golden.py`` ->
    ``tests/golden/*.npz``; checked by ``tests/test_oracle_golden.py``).

Parity status: renderer path PINNED by
the golden vectors G1-G12; its differentiable variant (row a19, the
fine-tune step's pass 2) PINNED by G11 (the reference's own autograd); the surface
renderer (``raycast.py``: row N4, models/ray_casting.py) PINNED by
``tests/golden/raycast_golden.npz`` (``make_golden_raycast.py``, checked by
``tests/test_oracle_raycast.py``).  CLIP ViT-B/32 (a23): "parity unpinned" - the
reference takes it from the un-vendored third-party ``clip`` package and
holds no test vectors for it; the oracle there is a restatement of the
published architecture cross-checked against ``transformers.CLIPVisionModel``.
"""
