"""Fine-tune step (row a19): the two-pass render / back-propagation of the reference Trainer
(models/frameworks/volsdf.py:689-783, :878-939; neus.py:455-576, :629-690).

    pass 1  (no grad)  whole image through the HIP renderer                       -> rgb [1, H*W, 3]
            style loss on the image (CLIP heads, criteria.py)                      -> d loss / d rgb
    pass 2  patches of `pass2_rays` rays (reference: 1200): HIP sampler (no grad, as volsdf.py:479), then the
            differentiable per-sample evaluation + compositing (autodiff.py),
            rgb_patch.backward(d loss / d rgb[patch]); eikonal = w * MSE(|nabla|, 1) over the patch's nablas, backward.
    caller  optimizer.step()  (train.py:247); with N ranks: dist.allreduce_gradients first.

The two passes and `perturb` (volsdf.py:724-728, :759-766, :982; neus.py:520-576, :742): the reference calls its renderer with
`render_kwargs_train` in BOTH passes, and `perturb` defaults to True there - pass 2 draws NEW uniform numbers for the 64 inverse-CDF
samples of every ray and back-propagates d loss / d rgb (evaluated on pass 1's image) through THOSE samples.  This Trainer does the
same: with perturb=True pass 1 keeps no per-point state, pass 2 gets fresh random samples (NeuS: the sampler runs again; VolSDF: see
`share_algorithm1` below) and the ray-level backward re-evaluates the per-point state there (`have_state = 0`).  With perturb=False re-sampling reproduces pass 1's samples exactly
(same weights, deterministic sampler), so pass 1 keeps its depths - and for VolSDF sdf / nablas / h7 - in HBM and pass 2 reads them:
the same numbers for one sampler and one forward evaluation less.  `Trainer(reuse_pass1_samples=True)` asks for that reuse under
perturb=True as well (a deviation: the gradient is then taken at the samples the loss was evaluated on);
`Trainer(resample_pass2=True / False)` forces either behaviour.  VolSDF: pass 2's samples come from pass 1's run of Algorithm 1
(`share_algorithm1`, render_two_draws: the rounds draw nothing and repeat exactly, only the final inverse-CDF reads uniform numbers).

Other differences from the reference, all deliberate: several reference patches share one launch group
(per-patch eikonal means kept); NeuS keeps `radiance_net` frozen exactly like neus.py:455-456.
"""
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import autodiff, hip
from . import dist as nd
from .nets import NeuS, VolSDF


_WARNED_AUTOGRAD = False


class Trainer(nn.Module):
    def __init__(self, model, w_eikonal: float = 0.1, use_eikonal: bool = True, pass2_rays: int = 1200, native: bool = None,
                 patches_per_launch: int = 4, freeze_radiance: bool = None, pass1_groups: int = 14, resample_pass2: bool = None,
                 reuse_pass1_samples: bool = False, share_algorithm1: bool = True):
        super().__init__()
        if not isinstance(model, (VolSDF, NeuS)):
            raise TypeError("Trainer expects a nerfart_amd VolSDF or NeuS model")
        self.model = model
        self.is_neus = isinstance(model, NeuS)
        self.w_eikonal, self.use_eikonal, self.pass2_rays = w_eikonal, use_eikonal, pass2_rays
        # native (None = True): pass 2 on the ray-level C entry points (csrc/render_backward.hip).  They read split-bf16 blobs, so a
        # training step on another precision is REFUSED (fp32-exact is an inference precision here) - checked when the step runs,
        # because set_precision may be called after get_model.  native=False is the torch-autograd formulation over library GEMMs:
        # the cross-check the native kernels are tested against, never chosen silently (it warns when it runs).
        self._native = native
        self._style_cfg = None
        # pass 2 draws its own samples (the reference's semantics, volsdf.py:759-766): None = whenever render_kwargs['perturb'] is
        # true (at perturb=False the sampler is deterministic and pass 1's samples ARE what pass 2 would draw: reused);
        # reuse_pass1_samples=True: the explicit opt-in to reuse under perturb=True as well
        if reuse_pass1_samples and resample_pass2:
            raise ValueError("Trainer: reuse_pass1_samples=True contradicts resample_pass2=True")
        self.resample_pass2, self.reuse_pass1_samples = resample_pass2, bool(reuse_pass1_samples)
        # VolSDF, pass 2 re-sampling: ONE run of Algorithm 1 serves both passes.  The weights do not change between the passes, so the
        # reference's second fine_sample call repeats the first one's rounds exactly (volsdf.py:159-285 draw nothing); only the final
        # opacity_invert_cdf_sample (:122-136, :287-300) reads fresh uniform numbers.  The sampler kernels invert a converged ray's CDF at
        # however many numbers they are given, so pass 1 asks for 2 x N_importance per ray: the first half are its own fine samples, the
        # second half pass 2's - bit-identical to a second sampler run with those numbers (tests), 0.37 s less per 480 x 270 step.
        # False: pass 2 runs the sampler again (the cross-check).  NeuS: every up-sampling round draws (neus.py:296): nothing to share.
        self.share_algorithm1 = bool(share_algorithm1)
        self._depths2 = None
        # the uniform numbers of perturb=True: None = torch.rand on the device; a callable (pass_no, first_ray, n_rays, n, device) ->
        # [n_rays, n] lets a test feed the draws the reference made (tests/golden/make_golden_finetune.py records them per pass)
        self.uniform_source = None
        self._global_rays = None          # (this rank's frame-ray indices, rays in the frame) while a ray-sharded step runs
        # native pass 2: this many of the reference's pass2_rays-ray patches share one set of kernel launches (the per-patch
        # eikonal means are kept); bounded by the kernels' 2^21 points per launch
        self.patches_per_launch = patches_per_launch
        # render_keep (VolSDF): pass 1 SAMPLES this many of pass 2's launch groups per set of sampler launches (its rounds each cost
        # a host read; results are chunk-invariant bit for bit); the per-point state is then evaluated group by group into tensors
        # of its own, which pass 2 releases as it consumes them (1 KiB per point: ~1 GB per 4 x 1200-ray group at P = 192).  14 groups =
        # 67,200 rays per sampler batch, about half the fused renderer's 131,072-ray chunk (measured: 4 -> 14 takes 8 ms off a 480 x 270 step)
        self.pass1_groups = max(1, pass1_groups)
        self._kept = None
        # neus.py:455-456: NeuS fine-tuning trains only the SDF net (and ln_s); pass freeze_radiance=False for the
        # reconstruction objective (reconstruction_step), which trains everything
        if self.is_neus if freeze_radiance is None else freeze_radiance:
            for p in model.radiance_net.parameters():
                p.requires_grad_(False)

    @property
    def native(self) -> bool:
        """Pass 2 on the hand-written kernels (the default).  The backward entry points and the state render_keep hands them read
        split-bf16 blobs only, so a model at another precision cannot train natively: that is an ERROR (no silent switch to library
        GEMMs), raised every time the flag is read.  Trainer(native=False) selects the autograd cross-check explicitly."""
        if self._native is False:
            global _WARNED_AUTOGRAD
            if not _WARNED_AUTOGRAD:
                _WARNED_AUTOGRAD = True
                warnings.warn("Trainer(native=False): pass 2 runs torch autograd over library GEMMs (the cross-check formulation), "
                              "not the hand-written kernels")
            return False
        if self.model.precision != "bf16x3":
            raise RuntimeError(f"training steps run on the split-bf16 kernels: call model.set_precision('bf16x3') (the model is at "
                               f"{self.model.precision!r}; 'fp32' is an inference precision - nerfart_*_render_bwd reads bf16x3 blobs). "
                               "Trainer(native=False) is the torch-autograd cross-check.")
        return True

    # ---- the losses of the fine-tune objective (the reference loads them in Trainer.__init__, volsdf.py:638-645) -----------------
    def configure_style_loss(self, args, target_hw):
        """Remember the config the style losses are built from; they are built on first use (a render-only run with an
        `is_finetune: True` YAML never loads CLIP / VGG - the reference does)."""
        self._style_cfg = (args, tuple(target_hw))

    def _ensure_style_loss(self):
        if getattr(self, "style_loss", None) is None and self._style_cfg is not None:
            from . import criteria
            args, hw = self._style_cfg
            self.style_loss = criteria.build_style_loss(args, hw, device=next(self.model.parameters()).device)
        return getattr(self, "style_loss", None)

    def _training_sampler(self):
        """A model whose sampler was calibrated for rendering (nets.calibrate_sampler: one-term weights error-compensated against ONE set of weights,
        ~2 s on the host) goes back to `mixed`'s hi + lo sampler before a step renders anything: the weights change every step."""
        m = self.model
        if getattr(m, "sampler_precision", None) == "fp16x1c":
            import warnings
            warnings.warn("Trainer: the model's sampler was calibrated for rendering (calibrate_sampler()); training uses the `mixed` mode's hi + lo "
                          "sampler - call calibrate_sampler() again before rendering with the trained weights")
            m.set_sampler_precision("fp16x2", guard=m.sampler_guard, late_round=m.sampler_late_round)

    def resamples(self, render_kwargs) -> bool:
        """Does pass 2 of a fine-tune step with these render kwargs run the sampler again (module docstring)?"""
        if self.reuse_pass1_samples:
            return False
        if self.resample_pass2 is not None:
            return bool(self.resample_pass2)
        return bool(render_kwargs.get("perturb", False))

    def shares_algorithm1(self, render_kwargs) -> bool:
        """Pass 2 re-samples AND takes its samples from pass 1's run of Algorithm 1 (render_two_draws): native VolSDF steps."""
        return bool(self.share_algorithm1 and self.resamples(render_kwargs) and not self.is_neus and self._native is not False)

    def _uniform(self, pass_no: int, first_ray: int, n_rays: int, n: int, device):
        if self.uniform_source is not None:
            if self._global_rays is not None:
                # ray-sharded step: this rank's rays are tiles of the frame, not a contiguous block - ask the source for the frame's rows and take
                # the ones this rank holds, so that a sharded step reads exactly the draws of the single-process step
                idx, n_frame = self._global_rays
                u = self.uniform_source(pass_no, 0, n_frame, n, device)
                if tuple(u.shape) != (n_frame, n):
                    raise ValueError(f"uniform_source returned {tuple(u.shape)}, expected {(n_frame, n)}")
                return u.to(device=device, dtype=torch.float32)[idx[first_ray:first_ray + n_rays]].contiguous()
            u = self.uniform_source(pass_no, first_ray, n_rays, n, device)
            if tuple(u.shape) != (n_rays, n):
                raise ValueError(f"uniform_source returned {tuple(u.shape)}, expected {(n_rays, n)}")
            return u.to(device=device, dtype=torch.float32).contiguous()
        return torch.rand(n_rays, n, device=device)

    # ---- pass 1 ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def render_image(self, render_fn, rays_o, rays_d, want_depths: bool = False, **render_kwargs):
        """Pass 1 on the fused renderer.  want_depths: also return the sample depths [N, P] of every ray (with perturb=False and
        unchanged weights pass 2 would re-derive exactly these - it can reuse them)."""
        self._training_sampler()
        kw = dict(render_kwargs)
        kw.pop("rayschunk", None)
        if kw.get("perturb", False) and self.uniform_source is not None:
            n = rays_o.reshape(-1, 3).shape[0]
            kw["uniforms"] = self._uniform(1, 0, n, kw.get("N_importance", 64), rays_o.device)
        rgb, depth, extras = render_fn(rays_o, rays_d, detailed_output=want_depths, require_nablas=True, calc_normal=True, **kw)
        if want_depths:
            d = extras["d_all" if self.is_neus else "d_vals"]
            return rgb, d.reshape(-1, d.shape[-1])
        return rgb

    # ---- pass 2 ---------------------------------------------------------------------------------------
    def _samples(self, o, dn, d_raw, rk, pass_no: int = 2, first_ray: int = 0):
        """Sample depths of a patch (no grad): the HIP sampler.  perturb=True draws the uniform numbers of the inverse-CDF
        samples from torch's generator (a fresh draw per call, as the reference's two passes draw separately)."""
        m = self.model
        surf_blob, rad_blob = m.packed()
        perturb = bool(rk.get("perturb", False))
        if self.is_neus:
            ni = rk.get("N_importance", 64)
            # the renderer's sampler on its own (its stage entry points: bit-identical depths, no frame rendered to get them)
            return hip.neus_sample(surf_blob, o, d_raw, obj_bounding_radius=rk.get("obj_bounding_radius", 1.0), n_samples=rk.get("N_samples", 64),
                                   n_importance=ni, n_upsample_iters=rk.get("N_upsample_iters", 4), precision=m.precision_id, **self._neus_algo(rk),
                                   u_new=self._uniform(pass_no, first_ray, o.shape[0], ni, o.device) if perturb else None)
        alpha, beta = m.forward_ab()
        ns, ni = rk.get("N_samples", 128), rk.get("N_importance", 64)
        near, far = rk.get("near", 0.0), rk.get("far", 6.0)
        # model.set_sampler_precision(...): Algorithm 1 on its own blob / precision (it carries no gradient, volsdf.py:479; the per-sample
        # state pass 2 differentiates is evaluated at the model's precision at whatever depths it returns)
        sa = m.sampler_args()
        d_fine, _, _ = hip.volsdf_fine_sample(sa["blob"], o, dn, near, far, rk.get("obj_bounding_radius", 3.0), float(alpha.detach()),
                                              float(beta.detach()), rk.get("epsilon", 0.1), 4 * ns, 4 * ns, ni,
                                              rk.get("max_upsample_steps", 5), rk.get("max_bisection_steps", 10),
                                              precision=sa["precision"], escalate=sa["escalate"], guard=sa["guard"], late_round=sa["late_round"],
                                              u_final=self._uniform(pass_no, first_ray, o.shape[0], ni, o.device) if perturb else None)
        t = hip.lin_table(ns, o.device)
        d_coarse = (near * (1.0 - t) + far * t)[None, :].expand(o.shape[0], ns)
        return torch.sort(torch.cat([d_coarse, d_fine], dim=-1), dim=-1)[0]

    @staticmethod
    def _neus_algo(rk) -> dict:
        """The up-sampling algorithm of the NeuS render kwargs (neus.py:733-736) in hip.neus_render's terms."""
        return dict(upsample_algo=rk.get("upsample_algo", "official_solution"), n_nograd_samples=rk.get("N_nograd_samples", 2048),
                    fixed_s_recp=rk.get("fixed_s_recp", 1 / 64.))

    def _launch_rays(self, P: int) -> int:
        k = max(1, min(self.patches_per_launch, (1 << 21) // max(self.pass2_rays * P, 1)))
        return self.pass2_rays * k

    @torch.no_grad()
    def render_keep(self, rays_o, rays_d, **rk):
        """Pass 1 of the native fine-tune step with state kept for pass 2.  VolSDF: on the per-stage C-ABI entries in the fused renderer's order (sampler,
        k_sdf_grad, radiance, composite), KEEPING every launch group's sample depths, sdf, nablas and layer-7 activations
        (1 KiB / point in HBM) for pass 2: the weights do not change between the passes and perturb=False is
        deterministic, so pass 2 would recompute exactly these.  Returns rgb [N, 3]; the state waits in self._kept."""
        self._training_sampler()
        m = self.model
        if not self.is_neus:
            from .volsdf import check_render_kwargs
            check_render_kwargs(**rk)                 # the staged pass 1 refuses what volume_render refuses
        if m.precision != "bf16x3":
            raise RuntimeError("render_keep keeps split-bf16 pass-1 state for the native pass 2: set_precision('bf16x3') first")
        o = rays_o.reshape(-1, 3).float().contiguous()
        d_raw = rays_d.reshape(-1, 3).float().contiguous()
        surf_blob, rad_blob = m.packed()
        if self.is_neus:
            # NeuS: the fused renderer's detailed outputs already hold what pass 2 needs at the P samples (depths, sdf,
            # nablas); the mid-point quantities are re-evaluated in pass 2
            ni = rk.get("N_importance", 64)
            P = rk.get("N_samples", 64) + ni
            step = self._launch_rays(P)
            s_val = float(m.forward_s().detach())
            kept, rgbs = [], []
            for i in range(0, o.shape[0], step):
                oi, di = o[i:i + step], d_raw[i:i + step]
                out = hip.neus_render(surf_blob, rad_blob, m.view_tiles, oi, di, obj_bounding_radius=rk.get("obj_bounding_radius", 1.0),
                                      s=s_val, n_samples=rk.get("N_samples", 64), n_importance=ni,
                                      n_upsample_iters=rk.get("N_upsample_iters", 4), white_bkgd=rk.get("white_bkgd", False),
                                      calc_normal=False, detailed=True, precision=m.precision_id, **self._neus_algo(rk),
                                      u_new=self._uniform(1, i, oi.shape[0], ni, o.device) if rk.get("perturb", False) else None)
                kept.append((out["d_all"], out["implicit_surface"].reshape(-1), out["implicit_nablas"].reshape(-1, 3), None))
                rgbs.append(out["rgb"])
            self._kept = kept
            return torch.cat(rgbs, 0) if rgbs else torch.zeros(0, 3, device=o.device)
        alpha, beta = m.forward_ab()
        ab = (float(alpha.detach()), float(beta.detach()))
        white = rk.get("white_bkgd", False)
        P = rk.get("N_samples", 128) + rk.get("N_importance", 64)
        step = self._launch_rays(P)
        big = step * self.pass1_groups
        kept, rgbs = [], []
        for i in range(0, o.shape[0], big):
            oi, di = o[i:i + big], d_raw[i:i + big]
            dn = hip.normalize_dirs(di)                                   # the kernel nerfart_volsdf_render_bwd normalises with: same points
            depths = self._samples(oi, dn, di, rk, 1, i).contiguous()     # one set of sampler launches for pass1_groups launch groups
            for j in range(0, oi.shape[0], step):                         # pass 2's launch groups: their OWN tensors, freed group by group
                dj = depths[j:j + step].contiguous()
                Rj = dj.shape[0]
                pts, v = hip.ray_points(oi[j:j + step], dn[j:j + step], dj)
                sdf, nab, h7 = hip.sdf_nabla_fwd(surf_blob, pts, m.obj_bounding_radius, precision=m.precision_id)
                rgb_pt = hip.radiance_fwd(rad_blob, m.view_tiles, pts, v, nab, h7, precision=m.precision_id)
                rgb, _, _ = hip.volsdf_composite(dj, sdf.reshape(Rj, P), rgb_pt.reshape(Rj, P, 3), ab[0], ab[1], white)
                kept.append((dj, sdf, nab, h7))
                rgbs.append(rgb)
        self._kept = kept
        return torch.cat(rgbs, 0) if rgbs else torch.zeros(0, 3, device=o.device)

    @torch.no_grad()
    def render_two_draws(self, rays_o, rays_d, **rk):
        """Pass 1 of a VolSDF fine-tune step whose pass 2 draws its own samples (perturb=True), with Algorithm 1 run ONCE: every batch of rays
        goes through the sampler with 2 x N_importance uniform numbers per ray - columns 0.. are pass 1's draw, N_importance.. pass 2's.
        Pass 1's image is evaluated at the first set (sdf + nabla, radiance, composite: the fused renderer's stages, nothing kept); the
        depths of the second set wait in self._depths2 [N, P] for backward_patches(depths_all=...), whose nerfart_volsdf_render_bwd
        (have_state = 0) evaluates the per-point state there.  Returns rgb [N, 3]."""
        self._training_sampler()
        m = self.model
        if self.is_neus:
            raise RuntimeError("render_two_draws: VolSDF only (NeuS draws in every up-sampling round)")
        from .volsdf import check_render_kwargs
        check_render_kwargs(**rk)                     # the staged pass 1 refuses what volume_render refuses
        if m.precision != "bf16x3":
            raise RuntimeError("render_two_draws feeds the native pass 2: set_precision('mixed') or ('bf16x3') first")
        o = rays_o.reshape(-1, 3).float().contiguous()
        d_raw = rays_d.reshape(-1, 3).float().contiguous()
        surf_blob, rad_blob = m.packed()
        sa = m.sampler_args()
        alpha, beta = m.forward_ab()
        ab = (float(alpha.detach()), float(beta.detach()))
        white = rk.get("white_bkgd", False)
        ns, ni = rk.get("N_samples", 128), rk.get("N_importance", 64)
        near, far = rk.get("near", 0.0), rk.get("far", 6.0)
        P = ns + ni
        step = self._launch_rays(P)
        big = step * self.pass1_groups
        t = hip.lin_table(ns, o.device)
        rgbs, deps2 = [], []
        for i in range(0, o.shape[0], big):
            oi, di = o[i:i + big], d_raw[i:i + big]
            n = oi.shape[0]
            dn = hip.normalize_dirs(di)
            u = torch.cat([self._uniform(1, i, n, ni, o.device), self._uniform(2, i, n, ni, o.device)], dim=1).contiguous()
            d_fine, _, _ = hip.volsdf_fine_sample(sa["blob"], oi, dn, near, far, rk.get("obj_bounding_radius", 3.0), ab[0], ab[1], rk.get("epsilon", 0.1),
                                                  4 * ns, 4 * ns, 2 * ni, rk.get("max_upsample_steps", 5), rk.get("max_bisection_steps", 10),
                                                  precision=sa["precision"], escalate=sa["escalate"], guard=sa["guard"], late_round=sa["late_round"], u_final=u)
            d_coarse = (near * (1.0 - t) + far * t)[None, :].expand(n, ns)
            dep1 = torch.sort(torch.cat([d_coarse, d_fine[:, :ni]], dim=-1), dim=-1)[0]
            deps2.append(torch.sort(torch.cat([d_coarse, d_fine[:, ni:]], dim=-1), dim=-1)[0])
            for j in range(0, n, step):
                dj = dep1[j:j + step].contiguous()
                Rj = dj.shape[0]
                pts, v = hip.ray_points(oi[j:j + step], dn[j:j + step], dj)
                sdf, nab, h7 = hip.sdf_nabla_fwd(surf_blob, pts, m.obj_bounding_radius, precision=m.precision_id)
                rgb_pt = hip.radiance_fwd(rad_blob, m.view_tiles, pts, v, nab, h7, precision=m.precision_id)
                rgb, _, _ = hip.volsdf_composite(dj, sdf.reshape(Rj, P), rgb_pt.reshape(Rj, P, 3), ab[0], ab[1], white)
                rgbs.append(rgb)
                del sdf, nab, h7, rgb_pt
        self._depths2 = torch.cat(deps2, 0) if deps2 else torch.zeros(0, P, device=o.device)
        return torch.cat(rgbs, 0) if rgbs else torch.zeros(0, 3, device=o.device)

    def backward_patches(self, rays_o, rays_d, gradient, depths_all=None, kept=None, **render_kwargs):
        """Pass 2: accumulates parameter gradients for d loss / d rgb = `gradient` [N, 3] (+ the eikonal term).
        depths_all [N, P]: sample depths from pass 1 (skips the re-sampling); kept: render_keep's per-group state of
        exactly these rays (skips the SDF re-evaluation too).  Returns the mean eikonal loss over the reference's
        pass2_rays-ray patches (what the reference prints)."""
        self._training_sampler()
        o_all = rays_o.reshape(-1, 3).float().contiguous()
        d_all_ = rays_d.reshape(-1, 3).float().contiguous()
        g_all = gradient.reshape(-1, 3)
        N = o_all.shape[0]
        white = render_kwargs.get("white_bkgd", False)
        eik_sum, n = 0.0, 0
        if self.native:
            accum = autodiff.GradAccumulator()                    # raw GEMM results summed over patches, flushed once
            ab = s_val = None
            if self.is_neus:
                s_val = float(self.model.forward_s().detach())
                P = render_kwargs.get("N_samples", 64) + render_kwargs.get("N_importance", 64)
            else:
                alpha, beta = self.model.forward_ab()
                ab = (float(alpha.detach()), float(beta.detach()))
                P = render_kwargs.get("N_samples", 128) + render_kwargs.get("N_importance", 64)
            if kept is not None:
                bounds, i = [], 0
                for grp in kept:
                    bounds.append((i, i + grp[0].shape[0]))
                    i += grp[0].shape[0]
                if i != N:
                    raise ValueError("backward_patches: the kept pass-1 state does not cover these rays")
            else:
                step = self._launch_rays(P)
                bounds = [(i, min(i + step, N)) for i in range(0, N, step)]
            fresh, fresh_lo = None, 0                              # re-sampled depths of the current batch of launch groups
            for gi, (i0, i1) in enumerate(bounds):
                o, d_raw, g = o_all[i0:i1], d_all_[i0:i1], g_all[i0:i1]
                state = None
                if kept is not None:
                    depths, state = kept[gi][0], kept[gi][1:]
                    kept[gi] = None                                # release the group's state as soon as it is consumed
                elif depths_all is not None:
                    depths = depths_all[i0:i1].contiguous()
                else:
                    # pass 2 draws its own samples (volsdf.py:759-766): ONE set of sampler launches per pass1_groups launch groups
                    # (the sampler's rounds each cost a host read; its results do not depend on how rays are chunked)
                    if fresh is None or i1 > fresh_lo + fresh.shape[0]:
                        j1 = bounds[min(gi + self.pass1_groups, len(bounds)) - 1][1]
                        with torch.no_grad():
                            fresh = self._samples(o_all[i0:j1], hip.normalize_dirs(d_all_[i0:j1]), d_all_[i0:j1], render_kwargs, 2, i0)
                        fresh_lo = i0
                    depths = fresh[i0 - fresh_lo:i1 - fresh_lo].contiguous()
                if self.is_neus:
                    eik = autodiff.neus_backward_samples_native(self.model, o, d_raw, depths, g, self.w_eikonal, self.use_eikonal, white,
                                                                s_val=s_val, accum=accum, eik_group_rays=self.pass2_rays, state=state)
                else:
                    eik = autodiff.volsdf_backward_samples_native(self.model, o, d_raw, depths, g, self.w_eikonal, self.use_eikonal, white,
                                                                  ab=ab, accum=accum, state=state, eik_group_rays=self.pass2_rays)
                eik_sum = eik_sum + eik
                n += -(-(i1 - i0) // self.pass2_rays)
                del state
            accum.flush(self.model)
            return float(eik_sum) / max(n, 1)
        for i in range(0, N, self.pass2_rays):
            o, d_raw = o_all[i:i + self.pass2_rays], d_all_[i:i + self.pass2_rays]
            dn = F.normalize(d_raw, dim=-1)
            with torch.no_grad():
                depths = self._samples(o, dn, d_raw, render_kwargs, 2, i) if depths_all is None else depths_all[i:i + self.pass2_rays].contiguous()
            fn = autodiff.neus_render_samples if self.is_neus else autodiff.volsdf_render_samples
            out = fn(self.model, o, dn, depths, white_bkgd=white)
            if self.use_eikonal:
                # the reference calls rgb.backward(g, retain_graph=True) and then eikonal.backward(): the same sum of
                # gradients from ONE traversal of the (double-backward) graph of the SDF net
                nn_ = out["implicit_nablas"].reshape(-1, 3).norm(dim=-1)
                eik = self.w_eikonal * F.mse_loss(nn_, torch.ones_like(nn_), reduction="mean")
                torch.autograd.backward([out["rgb"], eik], [g_all[i:i + self.pass2_rays], torch.ones_like(eik)])
                eik_sum += float(eik.detach())
            else:
                out["rgb"].backward(g_all[i:i + self.pass2_rays])
            n += 1
            del out
        return float(eik_sum) / max(n, 1)

    # ---- reconstruction-training branch (volsdf.py:784-824) -------------------------------------
    def reconstruction_step(self, render_fn, rays_o, rays_d, target_rgb, eikonal_points=None, w_eikonal: float = 0.1, optimizer=None,
                            mask_ignore=None, target_mask=None, w_mask: float = 0.0, **render_kwargs):
        """One step of the reconstruction objective on a batch of rays [N, 3].  Accumulates .grad; returns the losses.

        VolSDF (volsdf.py:784-824): mean |rgb - target| + w_eikonal * MSE(|nabla|, 1) over two nablas per ray - the sample
        of largest visibility weight and `eikonal_points` [N, 3] (the reference draws them uniformly in the bounding box,
        volsdf.py:799-801; here the caller does, so that runs are reproducible).
        NeuS (neus.py:578-617): |rgb - target| (masked mean over target_mask if given: `with_mask`) + w_eikonal * MSE over
        the nablas of ALL samples + w_mask * BCE(clamp(mask_volume, 1e-3, 1 - 1e-3), target_mask); the radiance net trains
        if its parameters require grad."""
        self._training_sampler()
        if self.is_neus:
            return self._reconstruction_step_neus(render_fn, rays_o, rays_d, target_rgb, w_eikonal, optimizer, mask_ignore, target_mask,
                                                  w_mask, **render_kwargs)
        if eikonal_points is None:
            raise ValueError("reconstruction_step (VolSDF) needs eikonal_points [N, 3]")
        m = self.model
        o = rays_o.reshape(-1, 3).float().contiguous()
        d = rays_d.reshape(-1, 3).float().contiguous()
        N = o.shape[0]
        kw = {k: v for k, v in render_kwargs.items() if k != "rayschunk"}
        with torch.no_grad():
            rgb, _, ex = render_fn(o[None], d[None], detailed_output=True, require_nablas=True, calc_normal=True, **kw)
            rgb = rgb.reshape(N, 3)
            depths = ex["d_vals"].reshape(N, -1)
            P = depths.shape[1]
            nab = ex["implicit_nablas"].reshape(N, P, 3)
            ind = ex["visibility_weights"].reshape(N, P - 1).argmax(dim=-1)                     # [N]
            n_sel = nab[torch.arange(N, device=o.device), ind]                                    # [N, 3]
            surf_blob, _ = m.packed()
            _, n_eik, _ = hip.sdf_nabla_fwd(surf_blob, eikonal_points.reshape(-1, 3).float().contiguous(), 0.0, want_h7=False,
                                            precision=m.precision_id)
            # losses and their cotangents
            diff = rgb - target_rgb.reshape(N, 3)
            if mask_ignore is not None:
                mk = mask_ignore.reshape(N, 1).float()
                loss_img = (diff.abs() * mk).sum() / (mk.sum() + 1e-10)
                g_rgb = torch.sign(diff) * mk / (mk.sum() + 1e-10)
            else:
                loss_img = diff.abs().mean()
                g_rgb = torch.sign(diff) / diff.numel()
            nn_all = torch.cat([n_sel.norm(dim=-1), n_eik.norm(dim=-1)])
            loss_eik = w_eikonal * ((nn_all - 1.0) ** 2).mean()
            coef = w_eikonal * 2.0 / nn_all.numel()
            g_sel = coef * ((n_sel.norm(dim=-1) - 1.0) / n_sel.norm(dim=-1))[:, None] * n_sel
            g_eik = coef * ((n_eik.norm(dim=-1) - 1.0) / n_eik.norm(dim=-1))[:, None] * n_eik
            nbar_extra = torch.zeros(N, P, 3, device=o.device)
            nbar_extra[torch.arange(N, device=o.device), ind] = g_sel
        if optimizer is not None:
            optimizer.zero_grad()
        dn = F.normalize(d, dim=-1)
        if self.native:
            alpha, beta = m.forward_ab()
            ab = (float(alpha.detach()), float(beta.detach()))
            accum = autodiff.GradAccumulator()
            for i in range(0, N, self.pass2_rays):
                sl = slice(i, i + self.pass2_rays)
                autodiff.volsdf_backward_samples_native(m, o[sl], d[sl], depths[sl].contiguous(), g_rgb[sl], use_eikonal=False,
                                                        white_bkgd=kw.get("white_bkgd", False), ab=ab, nbar_extra=nbar_extra[sl],
                                                        accum=accum)
            with torch.no_grad():
                autodiff.surface_param_backward(m, eikonal_points.reshape(-1, 3).float().contiguous(), g_eik.contiguous(), accum=accum)
            accum.flush(m)
        else:
            for i in range(0, N, self.pass2_rays):
                sl = slice(i, i + self.pass2_rays)
                out = autodiff.volsdf_render_samples(m, o[sl], dn[sl], depths[sl].contiguous(), white_bkgd=kw.get("white_bkgd", False))
                torch.autograd.backward([out["rgb"], out["implicit_nablas"]], [g_rgb[sl], nbar_extra[sl]])
            _, nab_e, _ = autodiff.surface_forward_with_nablas(m.implicit_surface, eikonal_points.reshape(-1, 3).float())
            nab_e.backward(g_eik)
        return {"loss_img": float(loss_img), "loss_eikonal": float(loss_eik), "total": float(loss_img + loss_eik)}

    def _reconstruction_step_neus(self, render_fn, rays_o, rays_d, target_rgb, w_eikonal, optimizer, mask_ignore, target_mask, w_mask,
                                  **render_kwargs):
        m = self.model
        o = rays_o.reshape(-1, 3).float().contiguous()
        d = rays_d.reshape(-1, 3).float().contiguous()
        N = o.shape[0]
        kw = {k: v for k, v in render_kwargs.items() if k != "rayschunk"}
        white = kw.get("white_bkgd", False)
        with torch.no_grad():
            rgb, _, ex = render_fn(o[None], d[None], detailed_output=True, calc_normal=False, **kw)
            rgb = rgb.reshape(N, 3)
            depths = ex["d_all"].reshape(N, -1).contiguous()
            P = depths.shape[1]
            acc = ex["mask_volume"].reshape(N)
            nn_ = ex["implicit_nablas"].reshape(N * P, 3).norm(dim=-1)
            loss_eik = w_eikonal * ((nn_ - 1.0) ** 2).mean()
            diff = rgb - target_rgb.reshape(N, 3)
            g_acc, loss_mask = None, torch.zeros((), device=o.device)
            weight = None                                             # per-ray weights of the masked image loss (neus.py:600-611)
            if target_mask is not None:
                tm = target_mask.reshape(N).float()
                mv = torch.clamp(acc, 1e-3, 1.0 - 1e-3)
                loss_mask = w_mask * F.binary_cross_entropy(mv, tm, reduction="mean")
                inside = ((acc > 1e-3) & (acc < 1.0 - 1e-3)).float()  # clamp passes no gradient outside
                g_acc = (w_mask / N) * (-(tm / mv) + (1.0 - tm) / (1.0 - mv)) * inside
                weight = tm if mask_ignore is None else tm * mask_ignore.reshape(N).float()
            elif mask_ignore is not None:
                weight = mask_ignore.reshape(N).float()
            if weight is not None:
                loss_img = (diff.abs() * weight[:, None]).sum() / (weight.sum() + 1e-10)
                g_rgb = torch.sign(diff) * weight[:, None] / (weight.sum() + 1e-10)
            else:
                loss_img = diff.abs().mean()
                g_rgb = torch.sign(diff) / diff.numel()
        if optimizer is not None:
            optimizer.zero_grad()
        dn = F.normalize(d, dim=-1)
        if self.native:
            if N * P > (1 << 21):
                raise ValueError("reconstruction_step (NeuS): at most 2^21 sample points per step (the eikonal mean spans the batch)")
            autodiff.neus_backward_samples_native(m, o, d, depths, g_rgb, w_eikonal, True, white, g_acc=g_acc)
        else:
            out = autodiff.neus_render_samples(m, o, dn, depths, white_bkgd=white, calc_normal=False)
            nn_g = out["implicit_nablas"].reshape(-1, 3).norm(dim=-1)
            eik = w_eikonal * F.mse_loss(nn_g, torch.ones_like(nn_g), reduction="mean")
            tensors, grads = [out["rgb"], eik], [g_rgb, torch.ones_like(eik)]
            if g_acc is not None:
                tensors.append(out["mask_volume"]); grads.append(g_acc)
            torch.autograd.backward(tensors, grads)
        out = {"loss_img": float(loss_img), "loss_eikonal": float(loss_eik), "total": float(loss_img + loss_eik + loss_mask)}
        if target_mask is not None:
            out["loss_mask"] = float(loss_mask)
        return out

    # ---- the reference's call shape (train.py:232) --------------------------------------------------------------
    def forward(self, args, indices, model_input, ground_truth, render_kwargs_train: dict, it: int, optimizer=None, render_fn=None,
                style_loss=None):
        """`trainer.forward(args, indices, model_input, ground_truth, render_kwargs_train, it, optimizer=optimizer)` as train.py
        calls it (volsdf.py:689-877, neus.py:458-627): rays of the batch's camera (all H x W in order when fine-tuning,
        args.data.N_rays random ones otherwise), the targets gathered at them, then

        * `args.training.is_finetune`: `finetune_step` - gradients are in `.grad` on return (the reference back-propagates
          inside forward too) and `losses` is the style loss (a tensor);
        * otherwise `reconstruction_step`.  train.py calls `optimizer.zero_grad(); losses['total'].backward()` AFTER forward, so
          the native pass 2 is deferred into that backward call (`_DeferredBackward`): `losses` holds loss_img, loss_eikonal
          (loss_mask), total as tensors, and `total.backward()` accumulates the gradients.

        Returns OrderedDict(losses=..., extras={'scalars': {...}, 'select_inds': ...}).  `render_fn` defaults to `self.render_fn`;
        `style_loss` to `self.style_loss`, which get_model configures from the YAML (built on first use: the reference builds its
        losses in __init__, volsdf.py:638-645)."""
        from collections import OrderedDict
        from . import rend_util
        render_fn = render_fn if render_fn is not None else getattr(self, "render_fn", None)
        style_loss = style_loss if style_loss is not None else (self._ensure_style_loss() if bool(args.training.is_finetune) else None)
        if render_fn is None:
            raise ValueError("Trainer.forward: set trainer.render_fn (the render_fn get_model returned) or pass render_fn=")
        dev = next(self.model.parameters()).device
        intrinsics, c2w = model_input["intrinsics"].to(dev), model_input["c2w"].to(dev)
        H, W = render_kwargs_train["H"], render_kwargs_train["W"]
        finetune = bool(args.training.is_finetune)
        rays_o, rays_d, select_inds = rend_util.get_rays(c2w, intrinsics, H, W, -1 if finetune else args.data.N_rays)
        target_rgb = torch.gather(ground_truth["rgb"].to(dev), 1, torch.stack(3 * [select_inds], -1))
        mask_ignore = torch.gather(model_input["mask_ignore"].to(dev), 1, select_inds) if "mask_ignore" in model_input else None
        rk = {k: v for k, v in render_kwargs_train.items() if k not in ("H", "W")}
        if finetune:
            if style_loss is None:
                raise ValueError("Trainer.forward (fine-tune): no style loss - get_model(args, [H, W]) configures it from the YAML when "
                                 "training.is_finetune; otherwise set trainer.style_loss (criteria.StyleLoss) or pass style_loss=")
            self.w_eikonal, self.use_eikonal = args.finetune.w_eikonal, bool(args.finetune.use_eikonal)
            out = self.finetune_step(render_fn, rays_o, rays_d, target_rgb, H, style_loss, optimizer=optimizer, **rk)
            losses = torch.tensor(out["loss"], device=dev)
        else:
            kwargs = dict(w_eikonal=args.training.w_eikonal, mask_ignore=mask_ignore)
            if self.is_neus:
                if args.training.get("with_mask", False):
                    kwargs.update(target_mask=torch.gather(model_input["object_mask"].to(dev), 1, select_inds), w_mask=args.training.w_mask)
            else:
                R = args.model.obj_bounding_radius
                kwargs["eikonal_points"] = torch.empty(rays_o.shape[-2], 3, device=dev).uniform_(-R, R)       # volsdf.py:799-801
            losses = _DeferredBackward.run(self, render_fn, rays_o, rays_d, target_rgb, kwargs, rk)
        extras = {"select_inds": select_inds}
        if self.is_neus:
            extras["scalars"] = {"1/s": 1.0 / self.model.forward_s().data}
        else:
            alpha, beta = self.model.forward_ab()
            extras["scalars"] = {"beta": beta.data, "alpha": alpha.data}
        return OrderedDict([("losses", losses), ("extras", extras)])

    # ---- one fine-tune step ---------------------------------------------------------------------------
    def finetune_step(self, render_fn, rays_o, rays_d, target_rgb, H: int, style_loss, optimizer=None, tile: int = None,
                      **render_kwargs):
        """style_loss(rgb_pred [B,3,H,W], rgb_gt [B,3,H,W]) -> scalar.  Returns dict(loss, eikonal, rgb).

        With torch.distributed initialised (one process per GPU) the step is ray-parallel: pass 1
        renders this rank's tiles and all-gathers the image; every rank evaluates the (cheap, deterministic) style
        loss on the full image and keeps its own rays' d loss / d rgb; pass 2 runs on its own rays; one flat
        all-reduce(SUM) of the gradients before the caller's optimizer.step().
        tile (rays dealt to a rank at a time) defaults to pass2_rays: every rank then owns WHOLE patches of the single-process
        step's patch grid, the per-patch eikonal means coincide and the all-reduced gradients equal the single-GPU step's up to
        summation order."""
        self._training_sampler()
        sharded = nd.world_size() > 1
        tile = self.pass2_rays if tile is None else tile
        resample = self.resamples(render_kwargs)           # pass 2 draws its own samples (the reference under perturb=True)
        keep = self.native and not resample                # pass 1 keeps its per-point state for pass 2 (render_keep)
        share = self.shares_algorithm1(render_kwargs)      # ... from the SAME run of Algorithm 1 (VolSDF: render_two_draws)
        self._kept = self._depths2 = None
        if sharded:
            kw = {k: v for k, v in render_kwargs.items() if k != "rayschunk"}
            n_frame = rays_o.reshape(-1, 3).shape[0]
            self._global_rays = (nd.shard_plan(n_frame, tile, rays_o.device).idx, n_frame)
            if keep:
                def fn(ro, rd, **kw_):
                    r = self.render_keep(ro, rd, **kw_)
                    return r, None, {"rgb": r[None]}
            elif share:
                def fn(ro, rd, **kw_):
                    r = self.render_two_draws(ro, rd, **kw_)
                    return r, None, {"rgb": r[None]}
            else:
                def fn(ro, rd, **kw_):
                    kw_ = {k: v for k, v in kw_.items() if k not in ("detailed_output", "require_nablas", "calc_normal")}
                    r = self.render_image(render_fn, ro, rd, **kw_)
                    return r, None, {"rgb": r.reshape(1, -1, 3)}
            rgb = nd.render_sharded(fn, rays_o.reshape(1, -1, 3), rays_d.reshape(1, -1, 3), keys=("rgb",), tile=tile,
                                    detailed_output=False, require_nablas=True, calc_normal=True, **kw)["rgb"]
            depths_all = None
        elif keep:
            rgb, depths_all = self.render_keep(rays_o, rays_d, **render_kwargs), None
        elif share:
            rgb = self.render_two_draws(rays_o, rays_d, **render_kwargs)
            depths_all = self._depths2
        elif resample:
            rgb, depths_all = self.render_image(render_fn, rays_o, rays_d, **render_kwargs), None
        else:
            rgb, depths_all = self.render_image(render_fn, rays_o, rays_d, want_depths=True, **render_kwargs)
        rgb = rgb.detach().reshape(1, -1, 3).requires_grad_(True)
        W = rgb.shape[1] // H
        to_img = lambda t: t.reshape(t.shape[0], H, W, 3).permute(0, 3, 1, 2)          # "B (H W) C -> B C H W"
        loss = style_loss(to_img(rgb), to_img(target_rgb.reshape(1, -1, 3)))
        loss.backward()
        gradient = rgb.grad.detach()
        if optimizer is not None:
            optimizer.zero_grad()
        if sharded:
            idx = nd.shard_plan(rgb.shape[1], tile, rgb.device).idx          # cached per (frame size, tile, world)
            eik = self.backward_patches(rays_o.reshape(-1, 3)[idx], rays_d.reshape(-1, 3)[idx], gradient[0][idx], kept=self._kept,
                                        depths_all=self._depths2, **render_kwargs)
            # ranks that own fewer parameters' gradients than others (none here: every rank touches every tensor)
            for p in self.model.parameters():
                if p.requires_grad and p.grad is None:
                    p.grad = torch.zeros_like(p)
            nd.allreduce_gradients([p for p in self.model.parameters() if p.requires_grad])
        else:
            eik = self.backward_patches(rays_o, rays_d, gradient[0], depths_all=depths_all, kept=self._kept, **render_kwargs)
        self._kept = self._depths2 = self._global_rays = None
        return {"loss": float(loss.detach()), "eikonal": eik, "rgb": rgb.detach()}


class _DeferredBackward(torch.autograd.Function):
    """Hands the gradients of a reconstruction step to whoever back-propagates its total: the reference's train loop calls
    `optimizer.zero_grad(); losses['total'].backward()` AFTER trainer.forward (train.py:240-242), so the step (render, losses,
    native pass 2) runs once in `run`, its gradients are parked, and `total.backward()` adds them to `.grad`."""

    @staticmethod
    def run(trainer, render_fn, rays_o, rays_d, target_rgb, kwargs, rk):
        from collections import OrderedDict
        params = [p for p in trainer.model.parameters()]
        before = [p.grad for p in params]
        for p in params:
            p.grad = None
        parts = trainer.reconstruction_step(render_fn, rays_o, rays_d, target_rgb, **kwargs, **rk)
        step = [p.grad for p in params]
        for p, g in zip(params, before):                           # forward leaves .grad as it found it
            p.grad = g
        dev = rays_o.device
        hook = torch.zeros((), device=dev, requires_grad=True)
        total = _DeferredBackward.apply(hook, torch.tensor(parts["total"], device=dev), params, step)
        losses = OrderedDict((k, torch.tensor(v, device=dev)) for k, v in parts.items() if k != "total")
        losses["total"] = total
        return losses

    @staticmethod
    def forward(ctx, hook, value, params, step):
        ctx.params, ctx.step = params, step
        return value.clone()

    @staticmethod
    def backward(ctx, g_total):
        for p, g in zip(ctx.params, ctx.step):
            if g is not None:
                p.grad = g * g_total if p.grad is None else p.grad + g * g_total
        return None, None, None, None
