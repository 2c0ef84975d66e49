"""Golden vectors for the data-side rows, from the REAL reference on CPU:

    python tests/golden/make_golden_campath.py          -> tests/golden/campath_golden.npz

* C1: render.py's camera-path functions (view_matrix, poses_avg, c2w_track_spiral with args.rot_rad, the spiral
  set-up of main_function :323-330, smoothed_motion_interpolation) on the reference's own cameras.
* C2: the first 6 (world_mat, scale_mat) pairs of the reference's data/fangzhou_nature/cameras.npz - DATA, the input of
  the load_K_Rt_from_P property tests (cv2 is absent, so its outputs cannot be captured).
render.py imports imageio / open3d-free paths only after the stubs of make_golden.py are installed.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    mg.install_stubs()
    sys.path.insert(0, mg.REF)
    os.chdir(mg.REF)
    import render as ref_render
    import nerfart_amd  # noqa: F401
    from nerfart_amd import dataio
    out = {}
    cams = np.load(os.path.join(mg.REF, "data/fangzhou_nature/cameras.npz"))
    n = 6
    for i in range(n):
        out[f"C2_world_mat_{i}"] = cams[f"world_mat_{i}"]
        out[f"C2_scale_mat_{i}"] = cams[f"scale_mat_{i}"]
    # poses through this package's decomposition (the reference's needs cv2); the path functions under test are
    # the reference's, fed with them
    c2ws = np.stack([dataio.load_K_Rt_from_P((cams[f"world_mat_{i}"] @ cams[f"scale_mat_{i}"])[:3, :4])[1] for i in range(n)])
    out["C1_c2ws"] = c2ws
    out["C1_poses_avg"] = ref_render.poses_avg(c2ws)
    out["C1_view_matrix"] = ref_render.view_matrix(np.array([0.2, -0.1, 1.0]), np.array([0.0, -1.0, 0.1]), np.array([1.0, 2.0, 3.0]))
    ref_render.args = types.SimpleNamespace(rot_rad=0.3, rot_percentile=85)
    c2w_center = ref_render.poses_avg(c2ws)
    up = c2ws[:, :3, 1].sum(0)
    rads = np.percentile(np.abs(c2ws[:, :3, 3]), 85, 0)
    focus = np.mean(np.linalg.norm(c2ws[:, :3, 3], axis=-1))
    out["C1_spiral"] = np.stack(ref_render.c2w_track_spiral(c2w_center, up, rads, focus * 0.8, zrate=0.0, rots=1, N=12))
    out["C1_smooth_40"] = ref_render.smoothed_motion_interpolation(2.0, 40)
    out["C1_smooth_7"] = ref_render.smoothed_motion_interpolation(1.0, 7, uniform_proportion=0.5)
    np.savez_compressed(os.path.join(HERE, "campath_golden.npz"), **{k: np.asarray(v) for k, v in out.items()})
    print("wrote campath_golden.npz", {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
