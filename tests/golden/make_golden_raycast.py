"""Golden vectors of the reference's surface renderer: models/ray_casting.py root_finding_surface_points,
sphere_tracing_surface_points and surface_render, run on CPU in the build container on the synthetic VolSDF / NeuS scenes.

    python tests/golden/make_golden_raycast.py        -> tests/golden/raycast_golden.npz

Only inputs / outputs are stored; weights are regenerated from seeds (conftest.scene_state) and guarded by the checksums of
renderer_golden.npz.  (utils/mesh_util.extract_mesh cannot be captured: it calls np.int, removed from numpy >= 1.24, and
skimage's marching cubes, absent here - DESIGN.md section 6b.)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg          # noqa: E402  (stubs + reference import recipe)


def main():
    mg.install_stubs()
    sys.path.insert(0, mg.REF)
    os.chdir(mg.REF)
    from utils import io_util, rend_util
    from models.frameworks import get_model as ref_get_model
    from models import ray_casting as rc
    import nerfart_amd  # noqa: F401
    from nerfart_amd import scene, frameworks
    torch.set_num_threads(8)
    out = {}
    H, W = 10, 9
    for fw, yaml_name, beta, near, far in (("VolSDF", "volsdf_fangzhou_nature.yaml", 0.01, 0.0, 6.0), ("NeuS", "neus_fangzhou_vangogh.yaml", None, 0.5, 4.5)):
        cfg = io_util.load_yaml(os.path.join(mg.REF, "configs", yaml_name))
        cfg.device_ids = [0]
        cfg.training.is_finetune = False
        torch.manual_seed(0)
        ref_model, _, _, _, _ = ref_get_model(cfg, [480, 270])
        torch.manual_seed(0)
        mine, _, _, _, _ = frameworks.get_model(scene.synthetic_config(fw))
        sd = scene.perturb_state(mine.state_dict(), beta=beta, seed=1)
        ref_model.load_state_dict(sd)
        ref_model.eval()
        c2w, K = scene.camera(H, W, cam_dist=2.5 if fw == "VolSDF" else 2.0, focal_scale=1.6)
        o, d, _ = rend_util.get_rays(c2w[None], K[None], H, W)
        o, d = o.contiguous().clone(), d.contiguous().clone()          # rays_o is a stride-0 expand of the camera centre
        # one ray starting INSIDE the surface (depth must come out 0) and one pointing away (no hit)
        o[0, 0] = torch.tensor([0.0, 0.0, 0.05]); d[0, 1] = -d[0, 1]
        dn = torch.nn.functional.normalize(d, dim=-1)
        t = f"{fw}_"
        out[t + "rays_o"], out[t + "rays_d"], out[t + "near"], out[t + "far"] = o[0], d[0], np.float32(near), np.float32(far)
        for tau in (0.0, 0.02):
            dd, pt, mask, msc = rc.root_finding_surface_points(ref_model.implicit_surface, o.clone(), dn.clone(), near=near, far=far,
                                                                N_steps=256, logit_tau=tau, N_secant_steps=8, fill_inf=(tau == 0.0))
            k = t + f"root_tau{tau}_"
            out[k + "d"], out[k + "pt"], out[k + "mask"], out[k + "mask_sign_change"] = dd[0], pt[0], mask[0], msc[0]
        with torch.no_grad():
            dd, pt, mask = rc.sphere_tracing_surface_points(ref_model.implicit_surface, o.clone(), dn.clone(), near=near, far=far, N_iters=20)
        out[t + "sphere_d"], out[t + "sphere_pt"], out[t + "sphere_mask"] = dd[0], pt[0], mask[0]
        for algo, cfgs in (("root_finding", dict(near=near, far=far, N_steps=256, N_secant_steps=8)), ("sphere_tracing", dict(near=near, far=far, N_iters=20))):
            col, dep, ex = rc.surface_render(o.clone(), d.clone(), ref_model, calc_normal=True, rayschunk=37, ray_casting_algo=algo,
                                             ray_casting_cfgs=cfgs)
            k = t + f"render_{algo}_"
            out[k + "rgb"], out[k + "depth"] = col[0], dep[0]
            assert list(ex.keys()) == ["implicit_nablas", "mask_surface", "normals_surface"]
            for kk, v in ex.items():
                out[k + kk] = v[0]
        print(fw, "hits:", int(out[t + "root_tau0.0_mask"].sum()), "of", H * W, "| sphere-traced live:", int(out[t + "sphere_mask"].sum()))
    path = os.path.join(HERE, "raycast_golden.npz")
    np.savez_compressed(path, **mg.t2n(out))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
