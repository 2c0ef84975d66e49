"""The CPU oracle's rendering of 2,048 strided rays of 8 orbit views of the benchmark scene (480 x 270, 128 + 64 spp, beta 0.01): what
tests/test_gpu_configs.py::test_cfg2_mixed_mode_over_eight_orbit_views compares the shipped `mixed` mode and pure split-bf16 with.

    python tests/golden/make_oracle_views.py          -> tests/golden/oracle_views_golden.npz   (~30 s per view on 32 threads, minutes on 8)

Not a reference fixture: an ORACLE fixture - oracle/render.py's output kept so that an 8-view GPU test does not spend 4.5 minutes of host time per run.
The committed file was written on the GPU box (tools/guard_sweep.py --oracle-cache, same call); tests/test_oracle_golden.py::test_oracle_views_fixture
re-runs the oracle on a subset of every view's rays on whatever CPU the suite runs on, and the GPU test runs it live, in full, on the two views
bench.py samples (orbit poses 1 and 5)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
POSES = (0, 1, 5, 11, 23, 37, 53, 71)
H, W, N = 480, 270, 2048


def view_rays(pose, rays=None):
    """(sel, rays_o [n, 3], rays_d [n, 3]) of the strided sample of orbit pose `pose` (oracle.render.get_rays = utils/rend_util.py:112-165 on the CPU);
    rays: a sub-sample of the N strided rays (indices into them)."""
    from nerfart_amd import scene
    from oracle import render as orender
    c2w, K = scene.camera(H, W, angle=scene.spiral(90)[pose])
    o, d = orender.get_rays(c2w, K, H, W)
    sel = torch.arange(0, H * W, (H * W) // N)[:N]
    if rays is not None:
        sel = sel[rays]
    return sel, o[sel].contiguous(), d[sel].contiguous()


def scene_sd():
    from nerfart_amd import scene, frameworks
    torch.manual_seed(0)
    model, _, _, _, _ = frameworks.get_model(scene.synthetic_config("VolSDF"))
    return scene.perturb_state(model.state_dict(), beta=0.01, seed=1)


def render(sd, o, d):
    from oracle import render as orender
    with torch.no_grad():
        return orender.volsdf_render(sd, o, d, near=0.0, far=6.0, obj_bounding_radius=3.0, N_samples=128, max_upsample_steps=6, chunk=max(o.shape[0], 64))


def main():
    sd = scene_sd()
    out = {"poses": np.array(POSES), "rays": np.array(N), "frame": np.array([H, W])}
    for p in POSES:
        _, o, d = view_rays(p)
        ref = render(sd, o, d)
        out[f"pose{p}_rgb"], out[f"pose{p}_iter_usage"] = ref["rgb"].numpy(), ref["iter_usage"].numpy()
        print("pose", p, "never converged:", int((ref["iter_usage"] < 0).sum()), flush=True)
    np.savez_compressed(os.path.join(HERE, "oracle_views_golden.npz"), **out)


if __name__ == "__main__":
    main()
