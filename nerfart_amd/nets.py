"""Host-side model containers with the reference's checkpoint layout.

Mirrors the reference classes (models/base.py: ImplicitSurface :131-282, RadianceNet :312-391;
models/frameworks/volsdf.py: VolSDF :304-370; neus.py: NeuS :80-123) at the level the hot path
needs: same constructor arguments, same ``state_dict`` keys / shapes / order
(``ln_beta`` | ``ln_s``, ``implicit_surface.obj_bounding_size``,
``implicit_surface.surface_fc_layers.{i}.{bias,weight_g,weight_v}``,
``radiance_net.layers.{i}.{bias,weight_g,weight_v}``), same geometric initialisation drawn from
the same RNG calls, and the same query methods - ``forward``, ``forward_surface``,
``forward_surface_with_nablas``, ``forward_ab`` / ``forward_s`` - whose arithmetic runs in the HIP
library (csrc/mlp_chain.hip) on packed weights.  There is no eager-PyTorch compute path here.
"""
from __future__ import annotations


import numpy as np
import torch
import torch.nn as nn

from . import hip


# Guard band of the `mixed` mode's sampler (hip.volsdf_fine_sample / nerfart_volsdf_fine_sample_guarded): a ray whose maximum error bound lies within
# DEFAULT_SAMPLER_GUARD * eps of eps at a convergence check of Algorithm 1 (volsdf.py:162-163, :240-242), or that never converges (:294-300), is
# sampled again on the split-bf16 kernels.  0.005 is the smallest guard at which the mode reproduces pure split-bf16's outlier statistics against the
# CPU oracle on all 8 measured views (profiles/r08_guard_sweep*.json, tools/guard_sweep.py: rays past 1e-3 <= bf16x3's + 1 per view, identical counts
# among the oracle-converged rays; 1.5 - 2.3 % of the rays run twice, + 2 % of a frame); the never-converged rays carry most of it (guard -> 0: 1.1 %).
DEFAULT_SAMPLER_GUARD = 0.005
# Third rule of the guarded sampler (nerfart_volsdf_fine_sample_guarded2, second session of round 6): a ray still active after up-sampling round 3 is sampled again on
# the split-bf16 kernels too.  From round 4 on every round starts from a 10-step bisection for beta+ whose threshold decisions feed the next round's sampling
# density - the branch-sensitive 3.6 % of a frame, and every converged ray any cheap sampler moved past 5e-4 on the 8 measured views.  With it the mode's
# frame differs from the pure split-bf16 frame on 0 - 2 of 129,600 rays by more than 1e-3 (37 - 66 without it) and reproduces its outlier counts against the
# CPU oracle on all 8 views exactly (profiles/r10_guard_sweep_*); + 5 ms per frame on the 2-MFMA sampler.
DEFAULT_SAMPLER_LATE_ROUND = 3


def embed_dim(multires: int, c: int = 3) -> int:
    return c if multires < 0 else c * (1 + 2 * multires)


class WNLinear(nn.Module):
    """A weight-normed linear layer as it sits in a reference checkpoint: parameters
    ``bias[out]``, ``weight_g[out,1]``, ``weight_v[out,in]`` in that order
    (``nn.utils.weight_norm(nn.Linear)``, models/base.py:226-227, :365-366)."""

    def __init__(self, weight: torch.Tensor, bias: torch.Tensor):
        super().__init__()
        self.bias = nn.Parameter(bias.detach().clone())
        self.weight_g = nn.Parameter(weight.detach().norm(dim=1, keepdim=True))
        self.weight_v = nn.Parameter(weight.detach().clone())

    @property
    def in_features(self):
        return self.weight_v.shape[1]

    @property
    def out_features(self):
        return self.weight_v.shape[0]


class ImplicitSurface(nn.Module):
    """SDF MLP container (models/base.py:131-231).  Initialisation reproduces :207-224 call for call."""

    def __init__(self, W=256, D=8, skips=(4,), W_geo_feat=256, input_ch=3, radius_init=1.0,
                 obj_bounding_size=2.0, geometric_init=True, embed_multires=6, weight_norm=True,
                 use_siren=False):
        super().__init__()
        if use_siren or not weight_norm:
            raise NotImplementedError("SIREN / un-normed surfaces are outside the hot-path scope (SURVEY.md 2, row 19)")
        self.radius_init, self.D, self.W, self.W_geo_feat = radius_init, D, W, W_geo_feat
        self.skips = list(skips)
        self.embed_multires = embed_multires
        self.register_buffer("obj_bounding_size", torch.tensor([obj_bounding_size]).float())
        in0 = embed_dim(embed_multires, input_ch)
        layers = []
        for l in range(D + 1):
            if l == D:
                out_dim = 1 + W_geo_feat if W_geo_feat > 0 else 1
            elif (l + 1) in self.skips:
                out_dim = W - in0
            else:
                out_dim = W
            in_dim = in0 if l == 0 else W
            lin = nn.Linear(in_dim, out_dim)            # consumes the same RNG draws as the reference
            if geometric_init:
                if l == D:
                    nn.init.normal_(lin.weight, mean=np.sqrt(np.pi) / np.sqrt(in_dim), std=0.0001)
                    nn.init.constant_(lin.bias, -radius_init)
                elif embed_multires > 0 and l == 0:
                    nn.init.constant_(lin.bias, 0.0)
                    nn.init.constant_(lin.weight[:, 3:], 0.0)
                    nn.init.normal_(lin.weight[:, :3], 0.0, np.sqrt(2) / np.sqrt(out_dim))
                elif embed_multires > 0 and l in self.skips:
                    nn.init.constant_(lin.bias, 0.0)
                    nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
                    nn.init.constant_(lin.weight[:, -(in0 - 3):], 0.0)
                else:
                    nn.init.constant_(lin.bias, 0.0)
                    nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
            layers.append(WNLinear(lin.weight.data, lin.bias.data))
        self.surface_fc_layers = nn.ModuleList(layers)

    # ---- boundary B3 as the reference's consumers call it (mesh_util.extract_mesh :110, ray_casting.py :179) ----------
    def _model(self):
        owner = getattr(self, "_owner", None)
        owner = owner() if owner is not None else None
        if owner is None:
            raise RuntimeError("ImplicitSurface queries run on its VolSDF / NeuS model's packed weight blob: build the model first")
        return owner

    def forward(self, x: torch.Tensor, return_h: bool = False):
        """ImplicitSurface.forward (models/base.py:243-263): sdf [...] (no sphere clamp) and, with return_h, the geometry
        feature [..., W_geo_feat].  HIP kernels (K2 / K3a + nerfart_geometry_feature for the feature rows of the last linear layer)."""
        m = self._model()
        if not return_h:
            return m._surface_query(x, 0.0, False, False)
        sdf, _, h7 = m._surface_query(x, 0.0, True, True)
        return sdf, self._geometry_feature(h7).reshape(*x.shape[:-1], -1)

    def forward_with_nablas(self, x: torch.Tensor, has_grad_bak=None):
        """ImplicitSurface.forward_with_nablas (base.py:265-282) under no_grad: (sdf, nablas, geometry feature)."""
        m = self._model()
        sdf, nab, h7 = m._surface_query(x, 0.0, True, True)
        return sdf, nab, self._geometry_feature(h7).reshape(*x.shape[:-1], -1)

    def _geometry_feature(self, h7):
        """Rows 1.. of the last linear layer on h7 (models/base.py:258-262): nerfart_geometry_feature - weight_norm fold + fp32 MFMA GEMM in the HIP
        library (round 6; rounds 1-5 used a rocBLAS GEMM here, off the frame path)."""
        last = self.surface_fc_layers[self.D]
        return hip.geometry_feature(last.weight_g, last.weight_v, last.bias, h7)


class RadianceNet(nn.Module):
    """Radiance MLP container (models/base.py:312-370)."""

    def __init__(self, D=4, W=256, skips=(), W_geo_feat=256, embed_multires=-1, embed_multires_view=-1,
                 use_view_dirs=True, weight_norm=True, use_siren=False):
        super().__init__()
        if use_siren or not weight_norm or len(skips) or not use_view_dirs:
            raise NotImplementedError("radiance variants outside the four reference configs (SURVEY.md 2, row 19)")
        self.D, self.W = D, W
        self.embed_multires, self.embed_multires_view = embed_multires, embed_multires_view
        in0 = embed_dim(embed_multires) + embed_dim(embed_multires_view) + 3 + W_geo_feat
        layers = []
        for l in range(D + 1):
            lin = nn.Linear(in0 if l == 0 else W, 3 if l == D else W)
            layers.append(WNLinear(lin.weight.data, lin.bias.data))
        self.layers = nn.ModuleList(layers)


class _PackedModel(nn.Module):
    """Shared machinery: folded + packed weight blobs, refreshed when parameters change."""

    def _init_packing(self):
        s, r = self.implicit_surface, self.radiance_net
        if r.embed_multires != -1 or r.embed_multires_view not in (-1, 4):
            raise NotImplementedError("radiance embed_multires must be -1 and embed_multires_view in (-1, 4)")
        self.view_tiles = 1 if r.embed_multires_view == -1 else 3
        # the kernels and the blob packers are written for ONE architecture per net (the four shipped configs): refuse any other at construction,
        # not with a fault inside nerfart_pack_*_blob (hip.check_layers holds the tensors themselves to the library's dims again at pack time)
        g, v, b = self._surface_layers()
        hip.check_layers("ImplicitSurface", hip.pack_layer_dims(False, s.embed_multires), g, v, b)
        R = list(r.layers)
        hip.check_layers("RadianceNet", hip.pack_layer_dims(True, self.view_tiles), [l.weight_g for l in R], [l.weight_v for l in R], [l.bias for l in R])
        self._bind_owner()
        self._blobs = None
        self._blob_key = None
        self.precision = "fp32"
        self.sampler_precision = None
        self.sampler_guard = 0.0
        self.sampler_late_round = 0
        self._sampler_blob = None
        self.radiance_precision = None
        self._radiance_blob = None
        self.render_stats = None        # a dict here collects {'rays', 'escalated'} over volume_render calls (bench.py, tests)

    def _bind_owner(self):
        import weakref
        # not a submodule: the surface net queries through this model's blob
        object.__setattr__(self.implicit_surface, "_owner", weakref.ref(self))

    def __deepcopy__(self, memo):
        """copy.deepcopy treats the weakref as atomic: a copied surface net would keep querying the ORIGINAL model's weights.
        Copy everything else, then point the copy's surface net at the copy; packed blobs are rebuilt on first use."""
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k in ("_blobs", "_blob_key", "_sampler_blob", "_radiance_blob") else copy.deepcopy(v, memo)
        new._bind_owner()
        return new

    def set_precision(self, precision: str):
        """'fp32'  : v_mfma_f32_16x16x4_f32, exact fp32 FMA chains (157 TFLOP/s peak);
        'bf16x3': operands split hi+lo in bf16, 3 x v_mfma_f32_16x16x32_bf16 per k-step, fp32 accumulate
                  (~2^-16 relative per product, 5.3x the fp32 MFMA rate);
        'mixed' : THE SHIPPED DEFAULT of get_model (round 5) = 'bf16x3' for every value that reaches a pixel or carries a gradient (the 192 final
                  samples: sdf, nabla, radiance, compositing, pass 2) + VolSDF's Algorithm 1 (512 (1 + rounds) no-gradient SDF queries per ray,
                  volsdf.py:479) on the 2-MFMA kernels: set_precision('bf16x3').set_sampler_precision('fp16x2').  It passes every reference-golden
                  assertion pure bf16x3 passes (tests/test_gpu_bf16x3.py, test_gpu_configs.py: both parametrised over the sampler).  NeuS has
                  no such sampler: 'mixed' is 'bf16x3' there;
        'fp16x2': ONE fp16 activation term x fp16 hi+lo weights, 2 x v_mfma_f32_16x16x32_f16 (11-bit activations, TF32 class) for ALL
                  forward kernels: a measurement variant - inference only, never the default.
        A precision is the whole mode: it resets set_sampler_precision."""
        if precision == "mixed":
            self.precision = "bf16x3"
            self.sampler_precision = "fp16x2" if hasattr(self, "ln_beta") else None
            self.sampler_guard = DEFAULT_SAMPLER_GUARD if hasattr(self, "ln_beta") else 0.0
            self.sampler_late_round = DEFAULT_SAMPLER_LATE_ROUND if hasattr(self, "ln_beta") else 0
        elif precision in hip.PRECISIONS:
            self.precision = precision
            self.sampler_precision = None
            self.sampler_guard = 0.0
            self.sampler_late_round = 0
        else:
            raise ValueError(f"precision must be one of {list(hip.PRECISIONS) + ['mixed']}")
        self._blobs = None
        self._sampler_blob = None
        self.radiance_precision = None
        self._radiance_blob = None
        return self

    @property
    def mode(self) -> str:
        """'mixed' | 'fp32' | 'bf16x3' | 'fp16x2' | 'bf16x3+<sampler precision> sampler'."""
        if self.sampler_precision is None or self.sampler_precision == self.precision:
            return self.precision
        if (self.precision, self.sampler_precision) == ("bf16x3", "fp16x2") and self.sampler_guard > 0 and self.sampler_late_round == DEFAULT_SAMPLER_LATE_ROUND:
            return "mixed"
        if (self.precision, self.sampler_precision) == ("bf16x3", "fp16x1c") and self.sampler_guard > 0 and self.sampler_late_round == DEFAULT_SAMPLER_LATE_ROUND:
            return "mixed (calibrated sampler)"
        late = f", late round {self.sampler_late_round}" if self.sampler_late_round > 0 else ""
        return f"{self.precision}+{self.sampler_precision} sampler" + (f" (guard {self.sampler_guard:g}{late})" if self.sampler_guard > 0 else " (unguarded)")

    def calibrate_sampler(self):
        """VolSDF, weights that are not about to change (rendering: the one line INTEGRATION.md section A adds to render.py after load_state_dict): Algorithm 1's
        SDF queries on ONE matrix instruction per product (C-ABI precision 5) over one-term fp16 weights that are error-compensated against this model's own
        activations (nerfart_amd/calibrate.py: ~2 s on the host, once per set of weights, cached by parameter version), behind the same guard as `mixed`
        (marginal decisions, never-converged rays and rays still active after round 3 sampled again in split-bf16).  - 15 % of a frame against `mixed` with
        the statistics of pure split-bf16 on all 8 measured views.  Training keeps `mixed`'s hi + lo sampler: a Trainer switches back."""
        if not hasattr(self, "ln_beta"):
            return self                                 # NeuS has no such sampler
        if self.precision != "bf16x3":
            raise RuntimeError("calibrate_sampler() is the shipped `mixed` mode's rendering form: set_precision('mixed') first")
        return self.set_sampler_precision("fp16x1c", guard=DEFAULT_SAMPLER_GUARD, late_round=DEFAULT_SAMPLER_LATE_ROUND)

    def set_sampler_precision(self, precision, guard: float = None, late_round: int = 0):
        """VolSDF only: run Algorithm 1's SDF queries (512 (1 + rounds) per ray, no gradient, volsdf.py:479) at another precision than the 192 final
        samples - e.g. model.set_precision("bf16x3").set_sampler_precision("fp16x2") (= set_precision("mixed")): every number that reaches
        a pixel is computed in split-bf16, only WHERE the fine samples sit comes from the 2-MFMA kernels - and, with guard > 0, not even that for
        the rays whose convergence decision is marginal (max B within guard * eps of eps) or that never converge: those are sampled again at the
        model's precision.  guard None = DEFAULT_SAMPLER_GUARD (the shipped mode's), 0 = the unguarded measurement variant of round 5.
        precision None: the sampler uses the model's precision."""
        if precision is not None and precision not in hip.SAMPLER_PRECISIONS:
            raise ValueError(f"precision must be one of {list(hip.SAMPLER_PRECISIONS)} or None")
        self.sampler_precision = precision
        self.sampler_guard = 0.0 if precision is None else (DEFAULT_SAMPLER_GUARD if guard is None else float(guard))
        self.sampler_late_round = 0 if precision is None else int(late_round)      # > 0: rays still active after that round are escalated too
        self._sampler_blob = None
        return self

    def set_radiance_precision(self, precision):
        """VolSDF's renderer only (nerfart_volsdf_render_staged_fwd): the radiance net of the final samples at another precision than sdf + nabla
        there - e.g. "fp16x2" (one fp16 activation term x fp16 hi + lo weights, 2 MFMAs per product; SURVEY.md 7 measured plain bf16 on the
        radiance net alone at rgb 4e-5 with identical sampling).  Inference only: pass 2 reads the split-bf16 blob.  None: the model's."""
        if precision is not None and precision not in hip.PRECISIONS:
            raise ValueError(f"precision must be one of {list(hip.PRECISIONS)} or None")
        self.radiance_precision = precision
        self._radiance_blob = None
        return self

    def packed_radiance(self):
        """(radiance blob, precision id) when the radiance net renders at its own precision; None otherwise."""
        if self.radiance_precision is None or self.radiance_precision == self.precision:
            return None
        key = (self.radiance_precision,) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._radiance_blob is None or self._radiance_blob[0] != key:
            with torch.no_grad():
                blob = self._pack_radiance(self.radiance_precision)
            self._radiance_blob = (key, blob)
        return self._radiance_blob[1], hip.PRECISIONS[self.radiance_precision]

    def sampler_args(self) -> dict:
        """Keyword arguments of hip.volsdf_fine_sample for this model's sampler: (blob, precision) + the escalation arithmetic and guard."""
        surf_blob, _ = self.packed()
        samp = self.packed_sampler()
        if samp is None:
            return dict(blob=surf_blob, precision=self.precision_id, escalate=None, guard=0.0, late_round=0)
        return dict(blob=samp[0], precision=samp[1], escalate=(surf_blob, self.precision_id), guard=self.sampler_guard, late_round=self.sampler_late_round)

    def _surface_layers(self):
        L = list(self.implicit_surface.surface_fc_layers)
        return [l.weight_g for l in L], [l.weight_v for l in L], [l.bias for l in L]

    def _pack_surface(self, precision: str) -> torch.Tensor:
        """The SDF net's blob at `precision` through the C ABI (nerfart_pack_surface_blob: weight_norm fold, unit-order permutation, hi / lo split
        on the device).  nerfart_amd/packing.py keeps the same layout as numpy plans - the source of truth of the CPU emulation
        (tests/emul_chain.py) - and tests/test_pack_plan.py holds the library's closed-form layout equal to them, entry for entry."""
        if precision in ("fp16x1c", "fp16x1"):
            # C-ABI precision 5, the 1-MFMA K2: one-term fp16 weights in the kernel's scaled activation recursion - "fp16x1c": error-compensated against this
            # model's own activations (pack-time calibration, what calibrate_sampler() ships); "fp16x1": rounded to nearest (measured, not shipped)
            from . import calibrate
            g, v, b, stats = calibrate.compensated_surface_layers(self, compensate=precision == "fp16x1c")
            if precision == "fp16x1c":
                self.calibration_stats = stats
        else:
            g, v, b = self._surface_layers()
        return hip.pack_surface_blob(hip.SAMPLER_PRECISIONS[precision], self.implicit_surface.embed_multires, g, v, b)

    def _pack_radiance(self, precision: str) -> torch.Tensor:
        last = self.implicit_surface.surface_fc_layers[self.implicit_surface.D]
        R = list(self.radiance_net.layers)
        return hip.pack_radiance_blob(hip.PRECISIONS[precision], self.view_tiles, (last.weight_g, last.weight_v, last.bias),
                                      [l.weight_g for l in R], [l.weight_v for l in R], [l.bias for l in R])

    def packed_sampler(self):
        """(surface blob, precision id) for the sampler when it runs at its own precision; None otherwise."""
        if self.sampler_precision is None or self.sampler_precision == self.precision:
            return None
        key = (self.sampler_precision,) + tuple((p.data_ptr(), p._version) for p in self.implicit_surface.parameters())
        if self._sampler_blob is None or self._sampler_blob[0] != key:
            with torch.no_grad():
                blob = self._pack_surface(self.sampler_precision)
            self._sampler_blob = (key, blob)
        return self._sampler_blob[1], hip.SAMPLER_PRECISIONS[self.sampler_precision]

    @property
    def precision_id(self) -> int:
        return hip.PRECISIONS[self.precision]

    def _param_key(self):
        return (self.precision,) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def packed(self):
        """(surface_blob, radiance_blob) for the current parameters; re-packed only after an
        in-place update (optimizer step / load_state_dict) - the reference re-folds weight_norm on
        every forward, 77x per ray chunk.  Two kernel launches per blob (pack_blob.hip)."""
        key = self._param_key()
        if self._blobs is None or key != self._blob_key:
            with torch.no_grad():
                self._blobs = (self._pack_surface(self.precision), self._pack_radiance(self.precision))
            self._blob_key = key
        return self._blobs

    # ---- point queries (boundaries B2 / B3 of SURVEY.md 8b) -------------------------------
    def _flat(self, x):
        return x.reshape(-1, 3).contiguous().float()

    def _surface_query(self, x, R_bg, with_nablas, want_h7):
        blob, _ = self.packed()
        xf = self._flat(x)
        if with_nablas:
            sdf, nab, h7 = hip.sdf_nabla_fwd(blob, xf, R_bg, want_h7=want_h7, precision=self.precision_id)
            return sdf.reshape(x.shape[:-1]), nab.reshape(x.shape), h7
        sdf = hip.sdf_fwd(blob, xf, R_bg, precision=self.precision_id)
        return sdf.reshape(x.shape[:-1])

    def _radiance_query(self, x, view_dirs, nablas, h7):
        _, rblob = self.packed()
        rgb = hip.radiance_fwd(rblob, self.view_tiles, self._flat(x), self._flat(view_dirs),
                               self._flat(nablas), h7, precision=self.precision_id)
        return rgb.reshape(x.shape)


class VolSDF(_PackedModel):
    """(models/frameworks/volsdf.py:304-370)"""

    def __init__(self, beta_init=0.1, speed_factor=1.0, input_ch=3, W_geo_feat=-1, obj_bounding_radius=3.0,
                 use_nerfplusplus=False, surface_cfg=None, radiance_cfg=None):
        super().__init__()
        if use_nerfplusplus:
            raise NotImplementedError("outside_scene: nerf++ is outside the hot-path scope (SURVEY.md 2, row 19)")
        self.speed_factor = speed_factor
        self.ln_beta = nn.Parameter(torch.Tensor([np.log(beta_init) / speed_factor]))
        self.use_sphere_bg = True
        self.obj_bounding_radius = obj_bounding_radius
        self.implicit_surface = ImplicitSurface(W_geo_feat=W_geo_feat, input_ch=input_ch,
                                                obj_bounding_size=obj_bounding_radius, **(surface_cfg or {}))
        if W_geo_feat < 0:
            W_geo_feat = self.implicit_surface.W
        self.radiance_net = RadianceNet(W_geo_feat=W_geo_feat, **(radiance_cfg or {}))
        self._init_packing()

    def forward_ab(self):
        beta = torch.exp(self.ln_beta * self.speed_factor)
        return 1.0 / beta, beta

    def forward_surface(self, x: torch.Tensor):
        """sdf = min(net(x), R - |x|) (volsdf.py:341-347).  The reference also returns the 256-d
        feature and drops it at every call site on this path; it is not
        materialised here - the second tuple element is None."""
        return self._surface_query(x, self.obj_bounding_radius, False, False), None

    def forward_surface_with_nablas(self, x: torch.Tensor):
        sdf, nab, h7 = self._surface_query(x, self.obj_bounding_radius, True, True)
        return sdf, nab, h7

    def forward(self, x: torch.Tensor, view_dirs: torch.Tensor = None, return_nablas=False):
        """(radiance, sdf, nablas) (volsdf.py:359-370)."""
        if view_dirs is None:
            raise NotImplementedError("use_view_dirs=False is not used by any reference config")
        sdf, nab, h7 = self._surface_query(x, self.obj_bounding_radius, True, True)
        rad = self._radiance_query(x, view_dirs, nab, h7)
        return rad, sdf, nab


class NeuS(_PackedModel):
    """(models/frameworks/neus.py:80-123)"""

    def __init__(self, variance_init=0.05, speed_factor=1.0, input_ch=3, W_geo_feat=-1, use_outside_nerf=False,
                 obj_bounding_radius=1.0, surface_cfg=None, radiance_cfg=None):
        super().__init__()
        if use_outside_nerf:
            raise NotImplementedError("NeRF++ outside net is outside the hot-path scope (SURVEY.md 2, row 19)")
        self.ln_s = nn.Parameter(torch.Tensor([-np.log(variance_init) / speed_factor]))
        self.speed_factor = speed_factor
        self.obj_bounding_radius = obj_bounding_radius
        self.implicit_surface = ImplicitSurface(W_geo_feat=W_geo_feat, input_ch=input_ch,
                                                obj_bounding_size=obj_bounding_radius, **(surface_cfg or {}))
        if W_geo_feat < 0:
            W_geo_feat = self.implicit_surface.W
        self.radiance_net = RadianceNet(W_geo_feat=W_geo_feat, **(radiance_cfg or {}))
        self._init_packing()

    def forward_s(self):
        return torch.exp(self.ln_s * self.speed_factor)

    def forward_sdf(self, x):
        """implicit_surface.forward(x) (no sphere clamp; neus.py:277,299)."""
        return self._surface_query(x, 0.0, False, False)

    def forward_sdf_with_nablas(self, x):
        sdf, nab, _ = self._surface_query(x, 0.0, True, False)
        return sdf, nab

    def forward_radiance(self, x, view_dirs, return_nablas=False):
        _, nab, h7 = self._surface_query(x, 0.0, True, True)
        return self._radiance_query(x, view_dirs, nab, h7)

    def forward(self, x, view_dirs, return_nablas=False):
        sdf, nab, h7 = self._surface_query(x, 0.0, True, True)
        return self._radiance_query(x, view_dirs, nab, h7), sdf, nab
