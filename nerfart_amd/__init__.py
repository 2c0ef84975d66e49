"""nerfart_amd: MI355X-native hot path of cassiePython/NeRF-Art.

Scope: the VolSDF / NeuS volumetric renderer - ray generation,
error-bounded hierarchical sampling, positional encoding, the SDF and radiance MLPs with
SDF normals, sigma/alpha compositing - as hand-written HIP kernels for gfx950 behind a
C-ABI shared library (include/nerfart_hip.h), with a PyTorch-ROCm host that mirrors the
reference's call shapes (render_fn / model.forward / model.forward_surface), YAML configs
and checkpoint key layout.

The compute path is the HIP library only: there is no CPU or eager-PyTorch fallback, and
nothing in this package imports ``oracle/``.
"""
__version__ = "0.1.0"
