"""Golden vectors for perturb=True (sample_cdf / sample_pdf with det=False, rend_util.py:269-272, :306-307), by running
the REAL reference (/root/reference) on CPU in the build container:

    python tests/golden/make_golden_perturb.py          -> tests/golden/perturb_golden.npz

The reference draws its uniform numbers with torch.rand inside the samplers (one draw per converged subset in
VolSDF's fine_sample, one per up-sampling round in NeuS).  torch.rand is wrapped while the reference runs, the draws
are recorded, and re-assembled into ONE table per ray (u_final [R, 64] / u_new [R, 64]) - the form in which this
package's C ABI and its oracle take them.  Stored: inputs, those tables, the reference's outputs.  Same stubs and
the same seed-regenerated scenes as make_golden.py.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


class RandRecorder:
    def __init__(self):
        self.draws = []
        self._orig = torch.rand

    def __enter__(self):
        def rec(*a, **k):
            r = self._orig(*a, **k)
            self.draws.append(r.detach().clone())
            return r
        torch.rand = rec
        return self

    def __exit__(self, *e):
        torch.rand = self._orig


def main():
    mg.install_stubs()
    sys.path.insert(0, mg.REF)
    os.chdir(mg.REF)
    from utils import io_util, rend_util
    from models.frameworks import get_model as ref_get_model
    from models.frameworks import volsdf as ref_volsdf
    import nerfart_amd  # noqa: F401
    from nerfart_amd import scene, frameworks
    torch.set_num_threads(8)
    out = {}
    g = torch.Generator().manual_seed(4321)

    # ---- P1: the two samplers with det=False ------------------------------------------------------------------
    d = torch.sort(torch.rand(8, 40, generator=g) * 6.0, dim=-1)[0]
    w = torch.rand(8, 39, generator=g); w[1, 5:20] = 0.0; w[2] = 0.0
    cdf = torch.cumsum(w / (w.sum(-1, keepdim=True) + 1e-3), -1) * 0.9
    with RandRecorder() as rr:
        torch.manual_seed(7)
        s_pdf = rend_util.sample_pdf(d, w, 16, det=False)
        s_cdf = rend_util.sample_cdf(d, cdf, 16, det=False)
    assert len(rr.draws) == 2
    out.update(P1_bins=d, P1_w=w, P1_cdf=cdf, P1_u_pdf=rr.draws[0], P1_u_cdf=rr.draws[1], P1_pdf16=s_pdf, P1_cdf16=s_cdf)

    # ---- scenes (as make_golden.py) ----------------------------------------------------------------------------
    states = {}
    for fw, yaml_name in (("VolSDF", "volsdf_fangzhou_nature.yaml"), ("NeuS", "neus_fangzhou_vangogh.yaml")):
        cfg = io_util.load_yaml(os.path.join(mg.REF, "configs", yaml_name))
        cfg.device_ids = [0]
        cfg.training.is_finetune = False
        torch.manual_seed(0)
        ref_model, _, rk_train, rk_test, ref_render = ref_get_model(cfg, [480, 270])
        states[fw] = (ref_model, ref_render, rk_test)

    def load_scene(fw, beta):
        ref_model, ref_render, rk_test = states[fw]
        torch.manual_seed(0)
        mine, _, _, _, _ = frameworks.get_model(scene.synthetic_config(fw))
        sd = scene.perturb_state(mine.state_dict(), beta=beta, seed=1)
        ref_model.load_state_dict(sd)
        return sd, ref_model, ref_render, dict(rk_test)

    H = W = 8
    c2w_s, K_s = scene.camera(H, W)
    ro, rd, _ = rend_util.get_rays(c2w_s[None], K_s[None], H, W)
    rdn = torch.nn.functional.normalize(rd, dim=-1)
    R = H * W

    # ---- P2: VolSDF fine_sample(perturb=True) + P3: volume_render(perturb=True) at beta = 0.01 ----------------
    sd, ref_model, ref_render, rk = load_scene("VolSDF", 0.01)
    out["P2_state_sha256"] = np.array(mg.state_checksum(sd))
    alpha, bnet = ref_model.forward_ab()
    t = torch.linspace(0, 1, 512).float()
    d_init = 0.0 * (1 - t) + 6.0 * torch.ones(1, R, 1) * t

    def assemble_volsdf(draws, usage):
        """draws: one [n_k, 64] per converged subset in call order (round 0, 1, ..., then the never-converged rest)."""
        u = torch.zeros(R, 64)
        order = [k for k in sorted(set(usage.tolist()) - {-1.0})] + ([-1.0] if (usage == -1).any() else [])
        assert len(order) == len(draws), (order, [tuple(x.shape) for x in draws])
        for k, dr in zip(order, draws):
            m = usage == k
            assert int(m.sum()) == dr.reshape(-1, 64).shape[0], (k, int(m.sum()), tuple(dr.shape))
            u[m] = dr.reshape(-1, 64)
        return u

    with RandRecorder() as rr, torch.no_grad():
        torch.manual_seed(11)
        d_fine, beta_map, usage = ref_volsdf.fine_sample(ref_model.forward_surface, d_init, ro, rdn, alpha_net=alpha, beta_net=bnet,
                                                          far=6.0 * torch.ones(1, R, 1), eps=0.1, max_iter=6, max_bisection=10,
                                                          final_N_importance=64, perturb=True, N_up=512)
    out.update(P2_u_final=assemble_volsdf(rr.draws, usage[0]), P2_d_fine=d_fine[0], P2_beta_map=beta_map[0], P2_iter_usage=usage[0])
    print("P2 iter_usage", usage.unique(return_counts=True), "draws", [tuple(x.shape) for x in rr.draws])

    rk["perturb"] = True
    with RandRecorder() as rr, torch.no_grad():
        torch.manual_seed(12)
        rgb, depth, ex = ref_render(ro, rd, require_nablas=True, calc_normal=True, detailed_output=True, **rk)
    out["P3_u_final"] = assemble_volsdf(rr.draws, ex["iter_usage"][0])
    for k in ("rgb", "depth_volume", "d_vals", "iter_usage", "beta_map"):
        out[f"P3_{k}"] = ex[k][0]

    # ---- P4: NeuS volume_render(perturb=True) ------------------------------------------------------------------
    sd, neus_model, neus_render, nrk = load_scene("NeuS", None)
    nrk["perturb"] = True
    with RandRecorder() as rr, torch.no_grad():
        torch.manual_seed(13)
        rgb, depth, ex = neus_render(ro, rd, calc_normal=True, detailed_output=True, **nrk)
    assert len(rr.draws) == 4 and all(x.reshape(-1, 16).shape[0] == R for x in rr.draws), [tuple(x.shape) for x in rr.draws]
    out["P4_u_new"] = torch.cat([x.reshape(R, 16) for x in rr.draws], dim=-1)
    for k in ("rgb", "depth_volume", "d_final", "implicit_surface"):
        out[f"P4_{k}"] = ex[k][0]
    out.update(P_c2w=c2w_s, P_K=K_s, P_H=np.array(H), P_W=np.array(W))

    np.savez_compressed(os.path.join(HERE, "perturb_golden.npz"), **mg.t2n(out))
    print("wrote perturb_golden.npz", {k: np.asarray(v).shape for k, v in mg.t2n(out).items()})


if __name__ == "__main__":
    main()
