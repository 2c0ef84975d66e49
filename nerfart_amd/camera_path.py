"""Render-time camera paths (reference render.py:21-133, :323-330): what `render.py --camera_path spiral
--num_views N` feeds `get_rays` with.  OpenCV convention (camera looks down +z, up = -y).  Pinned against the
reference's own functions (tests/golden/make_golden_campath.py -> tests/test_dataio.py)."""
import math

import numpy as np

from .rend_util import look_at, normalize        # noqa: F401  (look_at is part of the reference's render.py surface too)


def view_matrix(forward, up, cam_location):
    """[4,4] camera-to-world from a viewing direction, an up hint and a position (render.py:25-37)."""
    z = normalize(forward)
    x = normalize(np.cross(up, z))
    y = normalize(np.cross(z, x))
    m = np.stack((x, y, z, cam_location), axis=-1)
    bottom = np.array([[0.0, 0.0, 0.0, 1.0]])
    if m.ndim > 2:
        bottom = np.tile(bottom, [m.shape[0], 1, 1])
    return np.concatenate((m, bottom), axis=-2)


def poses_avg(poses):
    """The reference's active variant (render.py:46-51): NOT an average - the first pose re-orthonormalised."""
    return view_matrix(poses[0, :3, 2], poses[0, :3, 1], poses[0, :3, 3])


def c2w_track_spiral(c2w, up_vec, rads, focus: float, zrate: float, rots: int, N: int, rot_rad: float, zdelta: float = 0.0):
    """N poses on a circle of radius rot_rad around the centre pose's position, in the world x-y plane, all looking at
    the point `focus` in front of the centre camera (render.py:66-110: the spiral radii `rads`, `zrate` and `rots` are
    computed but do not enter the poses - kept in the signature for call compatibility; `rot_rad` is the reference's
    `args.rot_rad`, read there from a module global)."""
    focus_in_world = np.dot(c2w[:3, :4], np.array([0.0, 0.0, focus, 1.0]))
    center = np.asarray(c2w[:3, 3]).reshape(3)
    out = []
    for theta in np.linspace(0.0, 2.0 * np.pi, N + 1)[:-1]:
        loc = np.array([center[0] + rot_rad * np.cos(theta), center[1] + rot_rad * np.sin(theta), center[2]])
        out.append(look_at(loc, focus_in_world, up=up_vec))
    return out


def smoothed_motion_interpolation(full_range, num_samples, uniform_proportion=1 / 3.0):
    """Ease-in / uniform / ease-out sample positions over [0, full_range] (render.py:113-132)."""
    n_acc = max(math.ceil(num_samples * (1.0 - uniform_proportion) / 2.0), 2)
    n_uni = max(math.ceil(num_samples * uniform_proportion), 2)
    velocity = np.arange(n_acc)
    angle = np.cumsum(velocity)
    ratio = full_range / (2.0 * angle.max() + velocity.max() * n_uni)
    acc = angle * ratio
    uni = np.linspace(acc.max(), full_range - acc.max(), n_uni + 2)[1:-1]
    return np.concatenate([acc, uni, full_range - np.flip(acc)])


def spiral_path(c2ws: np.ndarray, num_views: int, rot_percentile: float = 85.0, rot_rad: float = 0.3):
    """render.py:323-330 for `--camera_path spiral`: c2ws [n,4,4] (the dataset's c2w_all) -> list of num_views [4,4]."""
    c2w_center = poses_avg(c2ws)
    up = c2ws[:, :3, 1].sum(0)
    rads = np.percentile(np.abs(c2ws[:, :3, 3]), rot_percentile, 0)
    focus_distance = np.mean(np.linalg.norm(c2ws[:, :3, 3], axis=-1))
    return c2w_track_spiral(c2w_center, up, rads, focus_distance * 0.8, zrate=0.0, rots=1, N=num_views, rot_rad=rot_rad)


def render_intrinsics(dataset, H=None, W=None, H_scale=None, W_scale=None):
    """(intrinsics [4,4] tensor, H, W) for rendering at a resolution other than the dataset's (render.py:289-311): ONLY the
    principal point is rescaled - fx, fy stay, so a smaller H x W is a crop of the field of view, as in the reference."""
    _, model_input, _ = dataset[0]
    K = model_input["intrinsics"].clone()
    Ho, Wo = dataset.H, dataset.W
    if H is not None:
        K[1, 2] *= (H / dataset.H)
        Ho = H
    if H_scale is not None:
        Ho = int(dataset.H * H_scale)
        K[1, 2] *= (Ho / dataset.H)
    if W is not None:
        K[0, 2] *= (W / dataset.W)
        Wo = W
    if W_scale is not None:
        Wo = int(dataset.W * W_scale)
        K[0, 2] *= (Wo / dataset.W)
    return K, Ho, Wo
