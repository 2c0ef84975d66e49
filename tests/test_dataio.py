"""Data-side rows, CPU: camera decomposition, the scene dataset, render camera paths.
load_K_Rt_from_P is UNPINNED against cv2 (absent here): property tests on the reference's own cameras
(tests/golden/campath_golden.npz C2_*: data copied from data/fangzhou_nature/cameras.npz).  The camera-path functions
are pinned against the reference's render.py (C1_*)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "campath_golden.npz")


@pytest.fixture(scope="module")
def cg():
    z = np.load(GOLD)
    return {k: z[k] for k in z.files}


def test_load_K_Rt_from_P_properties(cg):
    from nerfart_amd.dataio import load_K_Rt_from_P
    for i in range(6):
        P = (cg[f"C2_world_mat_{i}"] @ cg[f"C2_scale_mat_{i}"])[:3, :4].astype(np.float64)
        K4, pose = load_K_Rt_from_P(P)
        K = K4[:3, :3]
        assert K4.shape == (4, 4) and pose.shape == (4, 4) and pose.dtype == np.float32
        assert abs(K[2, 2] - 1.0) < 1e-12 and np.all(np.diag(K) > 0) and abs(K[1, 0]) + abs(K[2, 0]) + abs(K[2, 1]) < 1e-9
        R = pose[:3, :3].astype(np.float64).T                    # world -> camera
        c = pose[:3, 3].astype(np.float64)
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-6)
        assert np.linalg.det(R) > 0.999
        Pr = K @ np.concatenate([R, -(R @ c)[:, None]], axis=1)   # K [R | -R c] is P up to scale
        s = P[2, 3] / Pr[2, 3] if abs(Pr[2, 3]) > 1e-9 else P[0, 0] / Pr[0, 0]
        np.testing.assert_allclose(Pr * s, P, rtol=2e-5, atol=2e-4)
        assert s > 0                                              # the camera looks down +z: no sign flip of P
        np.testing.assert_allclose(P @ np.append(c, 1.0), 0.0, atol=2e-3)     # the centre is P's null vector
    # a synthetic camera with skew round-trips exactly
    Kt = np.array([[700.0, 1.5, 310.0], [0, 650.0, 255.0], [0, 0, 1.0]])
    from scipy.spatial.transform import Rotation
    Rt = Rotation.from_euler("xyz", [0.3, -0.5, 0.2]).as_matrix()
    ct = np.array([0.4, -1.2, 2.0])
    K4, pose = load_K_Rt_from_P(3.7 * Kt @ np.concatenate([Rt, -(Rt @ ct)[:, None]], axis=1))
    np.testing.assert_allclose(K4[:3, :3], Kt, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(pose[:3, :3], Rt.T, atol=1e-6)
    np.testing.assert_allclose(pose[:3, 3], ct, atol=1e-6)


def test_projection_sign_and_batch(cg):
    """A projection matrix is defined up to sign: -P decomposes to the same proper rotation; the batched
    decomposition equals the per-camera one."""
    from nerfart_amd.dataio import load_K_Rt_from_P, decompose_projections
    Ps = np.stack([(cg[f"C2_world_mat_{i}"] @ cg[f"C2_scale_mat_{i}"])[:3, :4] for i in range(6)]).astype(np.float64)
    Kb, cb = decompose_projections(Ps)
    for i in range(6):
        K1, c1 = load_K_Rt_from_P(Ps[i])
        Kn, cn = load_K_Rt_from_P(-Ps[i])
        np.testing.assert_array_equal(K1, Kb[i]); np.testing.assert_array_equal(c1, cb[i])
        np.testing.assert_allclose(Kn, K1, rtol=1e-12, atol=1e-12); np.testing.assert_allclose(cn, c1, atol=1e-6)
        assert np.linalg.det(cn[:3, :3].astype(np.float64)) > 0.999


def _make_scene(tmp_path, n=3, H=12, W=8):
    from PIL import Image
    z = np.load(GOLD)
    os.makedirs(tmp_path / "images"); os.makedirs(tmp_path / "matte")
    rng = np.random.default_rng(0)
    cams, imgs = {}, []
    for i in range(n):
        img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
        imgs.append(img)
        Image.fromarray(img).save(tmp_path / "images" / f"{i + 1:06d}.png")
        m = np.zeros((H, W, 3), np.uint8); m[2:9, 1:6] = 255; m[0, 0] = (127, 127, 127); m[0, 1] = (128, 128, 128)
        Image.fromarray(m).save(tmp_path / "matte" / f"{i + 1:06d}.png")
        cams[f"world_mat_{i}"], cams[f"scale_mat_{i}"] = z[f"C2_world_mat_{i}"], z[f"C2_scale_mat_{i}"]
    (tmp_path / "images" / "colmap_output.txt").write_text("not an image")
    np.savez(tmp_path / "cameras.npz", **cams)
    return imgs


def test_scene_dataset_matches_reference_contract(tmp_path):
    from nerfart_amd import dataio
    from nerfart_amd.config import ConfigDict
    imgs = _make_scene(tmp_path)
    ds = dataio.SceneDataset(False, str(tmp_path), downscale=1, scale_radius=3.0)
    assert len(ds) == 3 and (ds.H, ds.W) == (12, 8)
    idx, sample, gt = ds[1]
    assert idx == 1 and set(sample) == {"object_mask", "intrinsics", "c2w"} and set(gt) == {"rgb"}
    assert gt["rgb"].shape == (96, 3) and gt["rgb"].dtype == torch.float32
    np.testing.assert_array_equal(gt["rgb"].numpy(), imgs[1].reshape(-1, 3).astype(np.float32) / 255.0)      # row-major h * W + w
    mask = sample["object_mask"].reshape(12, 8)
    assert mask.dtype == torch.bool and mask[2:9, 1:6].all() and int(mask.sum()) == 35 + 1                    # 128 > 127.5 > 127
    # the farthest camera sits at scale_radius / 1.1 (DTU.py:68-71)
    norms = [float(c[:3, 3].norm()) for c in ds.c2w_all]
    np.testing.assert_allclose(max(norms), 3.0 / 1.1, rtol=1e-5)
    assert torch.stack(ds.c2w_all, dim=0).shape == (3, 4, 4)                      # render.py:309
    gtp = ds.get_gt_pose(scaled=True)
    assert gtp.shape == (3, 4, 4)
    np.testing.assert_allclose(gtp[0, :3, :3].numpy(), ds.c2w_all[0][:3, :3].numpy(), atol=1e-6)
    # downscale 2: intrinsics (not the skew) halve, images are 2 x 2 means
    ds2 = dataio.SceneDataset(False, str(tmp_path), downscale=2, scale_radius=3.0)
    assert (ds2.H, ds2.W) == (6, 4)
    K1, K2 = ds.intrinsics_all[0], ds2.intrinsics_all[0]
    np.testing.assert_allclose(K2[[0, 1, 0, 1], [0, 1, 2, 2]].numpy(), K1[[0, 1, 0, 1], [0, 1, 2, 2]].numpy() / 2, rtol=1e-6)
    assert float(K2[0, 1]) == float(K1[0, 1])
    box = (imgs[0].astype(np.float32) / 255.0).reshape(6, 2, 4, 2, 3).mean(axis=(1, 3))
    np.testing.assert_allclose(ds2.rgb_images[0].numpy(), box.reshape(-1, 3), atol=1e-6)
    # collate_fn + DataLoader, as train.py builds it
    dl = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, collate_fn=ds.collate_fn)
    ind, mi, g = next(iter(dl))
    assert ind.tolist() == [0, 1] and mi["c2w"].shape == (2, 4, 4) and g["rgb"].shape == (2, 96, 3)
    # get_data from a config
    cfg = ConfigDict({"data": ConfigDict({"data_dir": str(tmp_path), "downscale": 1, "scale_radius": 3.0})})
    d1, d2 = dataio.get_data(cfg, return_val=True, val_downscale=2)
    assert (d1.H, d2.H) == (12, 6)
    with pytest.raises(NotImplementedError):
        dataio.get_data(ConfigDict({"data": ConfigDict({"type": "BlendedMVS", "data_dir": str(tmp_path), "downscale": 1})}))


def test_camera_paths_match_reference(cg):
    from nerfart_amd import camera_path as cp
    c2ws = cg["C1_c2ws"]
    np.testing.assert_allclose(cp.poses_avg(c2ws), cg["C1_poses_avg"], atol=1e-12)
    np.testing.assert_allclose(cp.view_matrix(np.array([0.2, -0.1, 1.0]), np.array([0.0, -1.0, 0.1]), np.array([1.0, 2.0, 3.0])),
                               cg["C1_view_matrix"], atol=1e-12)
    np.testing.assert_allclose(np.stack(cp.spiral_path(c2ws, 12, rot_percentile=85, rot_rad=0.3)), cg["C1_spiral"], atol=1e-12)
    np.testing.assert_allclose(cp.smoothed_motion_interpolation(2.0, 40), cg["C1_smooth_40"], atol=1e-12)
    np.testing.assert_allclose(cp.smoothed_motion_interpolation(1.0, 7, uniform_proportion=0.5), cg["C1_smooth_7"], atol=1e-12)
    # the poses are usable by get_rays: orthonormal, looking at the focus point
    for m in cp.spiral_path(c2ws, 5):
        np.testing.assert_allclose(m[:3, :3].T @ m[:3, :3], np.eye(3), atol=1e-6)


def test_rescale_equals_the_scipy_call_skimage_makes():
    """skimage 0.19.3 (the reference's pin, requirements.txt:43) implements `rescale(img, 1 / downscale, anti_aliasing=False)` for
    order 1 as `scipy.ndimage.zoom(img, out / in, order=1, mode='mirror', grid_mode=True)` (skimage/transform/_warps.py resize:
    default mode 'reflect' -> ndimage 'mirror'); scipy is present here, skimage is not - `_rescale` is held to that call."""
    import scipy.ndimage as ndi
    from nerfart_amd.dataio import _rescale
    rng = np.random.default_rng(0)
    for shape in ((12, 8, 3), (540, 960, 3), (37, 53)):
        img = rng.random(shape).astype(np.float32)
        for ds in (2, 4, 8, 1.5):
            mine = _rescale(img, ds)
            out = (int(round(shape[0] / ds)), int(round(shape[1] / ds)))
            zoom = [out[0] / shape[0], out[1] / shape[1]] + ([1] if len(shape) == 3 else [])
            ref = ndi.zoom(img, zoom, order=1, mode="mirror", grid_mode=True, prefilter=False)
            assert mine.shape == ref.shape
            exact = (shape[0] % out[0] == 0) and (shape[1] % out[1] == 0)          # otherwise scipy's float64 vs torch's float32 coordinates
            np.testing.assert_allclose(mine, ref, atol=5e-7 if exact else 5e-5, rtol=0)
