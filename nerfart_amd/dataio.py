"""Scene dataset + camera loading: the data formats on the input side of the renderer.

Mirrors the reference's `dataio.get_data` / `dataio/DTU.py:SceneDataset` (IDR-style scene folder: `images/`, `matte/`,
`cameras.npz` holding `world_mat_i`, `scale_mat_i`) and the helpers they use (`utils/io_util.py:19-56` glob_imgs /
load_rgb / load_mask, `utils/rend_util.py:8-25` load_K_Rt_from_P) - same names, arguments, attributes and return
values, so `render.py` / `train.py` style callers run unchanged.  The reference leans on cv2, imageio and skimage for
three things; none of them is in this image, so they are restated on numpy / scipy / PIL / torch:

* `cv2.decomposeProjectionMatrix` -> RQ decomposition (one batched QR over all views) with cv2's conventions (positive
  diagonal of K, proper rotation, camera centre = null vector of P).  PARITY UNPINNED against cv2 itself (absent); pinned by properties on the reference's own
  `data/fangzhou_nature/cameras.npz` cameras (tests/test_dataio.py: K [R | -R c] reproduces P, R orthonormal).
* `imageio.imread` -> PIL; `skimage.img_as_float32` -> / 255.
* `skimage.transform.rescale(img, 1 / downscale, anti_aliasing=False)` (order 1, half-pixel centres) ->
  `F.interpolate(mode="bilinear", align_corners=False, antialias=False)`, the same sampling rule (for an integer
  downscale every output pixel is the mean of a 2 x 2 neighbourhood of input pixels in both).  Pinned against the scipy call
  skimage 0.19.3 makes for it (`ndimage.zoom(order=1, mode="mirror", grid_mode=True)`, tests/test_dataio.py); skimage itself is absent.
"""
import glob
import os

import numpy as np
import torch
import torch.nn.functional as F


def glob_imgs(path):
    imgs = []
    for ext in ["*.png", "*.jpg", "*.JPEG", "*.JPG"]:
        imgs.extend(glob.glob(os.path.join(path, ext)))
    return imgs


def _rescale(img: np.ndarray, downscale: float) -> np.ndarray:
    """[H, W(, C)] float32 -> [round(H / downscale), round(W / downscale)(, C)], bilinear, no anti-aliasing."""
    t = torch.from_numpy(np.ascontiguousarray(img)).float()
    chw = t[None, None] if t.ndim == 2 else t.permute(2, 0, 1)[None]
    H, W = chw.shape[-2:]
    size = (int(round(H / downscale)), int(round(W / downscale)))
    out = F.interpolate(chw, size=size, mode="bilinear", align_corners=False, antialias=False)[0]
    return (out[0] if t.ndim == 2 else out.permute(1, 2, 0)).numpy()


def load_rgb(path, downscale=1):
    """[3, H, W] float32 in [0, 1] (io_util.py:37-47)."""
    from PIL import Image
    img = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0
    if downscale != 1:
        img = _rescale(img, downscale)
    return img.transpose(2, 0, 1)


def load_mask(path, downscale=1):
    """[H, W] bool: grey value > 127.5 (io_util.py:49-56; ITU-R 601 luma as imageio's as_gray)."""
    from PIL import Image
    rgb = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32)
    alpha = rgb[..., 0] * 0.299 + rgb[..., 1] * 0.587 + rgb[..., 2] * 0.114
    if downscale != 1:
        alpha = _rescale(alpha, downscale)
    return alpha > 127.5


def decompose_projections(P):
    """All cameras at once: P [n, 3, 4] = s K [R | -R c]  ->  (K [n, 4, 4] float64 normalised to K[2,2] = 1 with a positive
    diagonal, c2w [n, 4, 4] float32 = [R^T | c]).  What `cv2.decomposeProjectionMatrix` + the reference's
    load_K_Rt_from_P (rend_util.py:8-25) give per view.

    RQ through one batched QR: with J the row-reversal, (J M)^T = Q U  =>  M = (J U^T J)(J Q^T), J U^T J upper triangular.
    A projection matrix is defined up to sign; det(M) < 0 would make R a reflection, so P is negated first (cv2 returns
    a proper rotation in that case too)."""
    P = np.asarray(P, dtype=np.float64).reshape(-1, 3, 4)
    P = P * np.where(np.linalg.det(P[:, :, :3]) < 0, -1.0, 1.0)[:, None, None]
    M = P[:, :, :3]
    Q, U = np.linalg.qr(np.swapaxes(M[:, ::-1, :], 1, 2))
    K = np.swapaxes(U, 1, 2)[:, ::-1, ::-1]
    R = np.swapaxes(Q, 1, 2)[:, ::-1, :]
    sgn = np.where(np.diagonal(K, axis1=1, axis2=2) < 0, -1.0, 1.0)           # K diag(s) . diag(s) R = K R
    K, R = K * sgn[:, None, :], R * sgn[:, :, None]
    centre = -np.linalg.solve(M, P[:, :, 3:4])[:, :, 0]
    K4 = np.tile(np.eye(4), (P.shape[0], 1, 1))
    K4[:, :3, :3] = K / K[:, 2:3, 2:3]
    c2w = np.tile(np.eye(4, dtype=np.float32), (P.shape[0], 1, 1))
    c2w[:, :3, :3] = np.swapaxes(R, 1, 2)
    c2w[:, :3, 3] = centre
    return K4, c2w


def load_K_Rt_from_P(P):
    """One camera: P [3, 4] -> (intrinsics [4, 4] float64, pose [4, 4] float32 camera-to-world)  (rend_util.py:8-25)."""
    K4, c2w = decompose_projections(np.asarray(P)[None, :3, :4])
    return K4[0], c2w[0]


def _stack_cameras(cam_file, n, scaled=True):
    """world_mat_i (@ scale_mat_i) of the first n views of a cameras.npz as one [n, 3, 4] float32 array (one file read)."""
    with np.load(cam_file) as z:
        world = np.stack([z["world_mat_%d" % i] for i in range(n)]).astype(np.float32)
        if not scaled:
            return world[:, :3, :4]
        scale = np.stack([z["scale_mat_%d" % i] for i in range(n)]).astype(np.float32)
    return (world @ scale)[:, :3, :4]


class SceneDataset(torch.utils.data.Dataset):
    """IDR-style scene folder -> per-view items with the reference dataset's contract (dataio/DTU.py:11-155):
    `ds[i] = (i, {"object_mask" [H W] bool, "intrinsics" [4,4], "c2w" [4,4]}, {"rgb" [H W, 3]})`, attributes `n_images, H, W,
    downscale, instance_dir, cam_file, train_cameras, intrinsics_all, c2w_all, rgb_images, object_masks` (per-view lists, as
    render.py's `torch.stack(dataset.c2w_all)` expects - here views into one stacked tensor each).

    Cameras are decomposed for all views in one batched call; the intrinsics' focal lengths and principal point are
    divided by `downscale` (the skew, a ratio, is not: DTU.py:58-63) and, with scale_radius > 0, camera centres are
    scaled so that the farthest one sits at scale_radius / 1.1 (DTU.py:68-71)."""

    def __init__(self, train_cameras, data_dir, downscale=1., cam_file=None, scale_radius=-1):
        assert os.path.exists(data_dir), f"Data directory {data_dir} is empty"
        self.instance_dir, self.train_cameras, self.downscale = data_dir, train_cameras, downscale
        frames = sorted(glob_imgs(os.path.join(data_dir, "images")))
        mattes = sorted(glob_imgs(os.path.join(data_dir, "matte")))            # only the NeuS + mask objective reads them
        self.n_images = len(frames)
        self.cam_file = os.path.join(data_dir, cam_file if cam_file is not None else "cameras.npz")

        K4, c2w = decompose_projections(_stack_cameras(self.cam_file, self.n_images))
        K4[:, [0, 1, 0, 1], [0, 1, 2, 2]] /= downscale
        if scale_radius > 0:
            c2w[:, :3, 3] *= scale_radius / np.linalg.norm(c2w[:, :3, 3], axis=1).max() / 1.1
        self.intrinsics_all = list(torch.from_numpy(K4).float().unbind(0))
        self.c2w_all = list(torch.from_numpy(c2w).float().unbind(0))

        pixels = np.stack([load_rgb(f, downscale) for f in frames])           # [n, 3, H, W]
        self.H, self.W = pixels.shape[-2:]
        self.rgb_images = list(torch.from_numpy(np.ascontiguousarray(pixels.reshape(self.n_images, 3, -1).transpose(0, 2, 1))).float().unbind(0))
        if mattes and len(mattes) != self.n_images:
            raise ValueError(f"{len(mattes)} mattes for {self.n_images} images: the mask folder must hold one matte per image (or none)")
        self.has_mattes = bool(mattes)
        if mattes:
            self.object_masks = list(torch.from_numpy(np.stack([load_mask(m, downscale).reshape(-1) for m in mattes])).bool().unbind(0))
        else:
            # no matte folder: the reference leaves its list empty and __getitem__ fails; here every pixel counts as object, which
            # is only right for objectives WITHOUT a mask term (VolSDF; NeuS with w_mask = 0) - `has_mattes` lets the caller refuse
            import warnings
            warnings.warn("SceneDataset: no mattes found - object_mask is all ones; do not train a mask loss (NeuS w_mask > 0) on this scene")
            self.object_masks = list(torch.ones(self.n_images, self.H * self.W, dtype=torch.bool).unbind(0))

    def __len__(self):
        return self.n_images

    def __getitem__(self, idx):
        sample = {"object_mask": self.object_masks[idx], "intrinsics": self.intrinsics_all[idx]}
        if not self.train_cameras:
            sample["c2w"] = self.c2w_all[idx]
        return idx, sample, {"rgb": self.rgb_images[idx]}

    @staticmethod
    def collate_fn(batch):
        """[(idx, sample, truth), ...] -> (LongTensor [b], {key: [b, ...]}, {key: [b, ...]}) - the DataLoader hook train.py
        installs (DTU.py:111-127)."""
        idx, samples, truths = zip(*batch)

        def stacked(dicts):
            return {k: torch.stack([d[k] for d in dicts]) for k in dicts[0]}
        return torch.LongTensor(idx), stacked(samples), stacked(truths)

    def get_scale_mat(self):
        with np.load(self.cam_file) as z:
            return z["scale_mat_0"]

    def get_gt_pose(self, scaled=True):
        """[n, 4, 4] camera-to-world WITHOUT the scale_radius normalisation (DTU.py:132-147)."""
        return torch.from_numpy(decompose_projections(_stack_cameras(self.cam_file, self.n_images, scaled))[1]).float()


_LAYOUTS = {"DTU": SceneDataset}           # the folder layout every reference config uses; 'custom' / 'BlendedMVS' are out of scope


def get_data(args, return_val=False, val_downscale=4.0, **overwrite_cfgs):
    """Dataset(s) of a config (`dataio.get_data`, dataio/__init__.py:1-26): the training set and, with return_val, a second
    instance at `val_downscale`."""
    layout = args.data.get("type", "DTU")
    if layout not in _LAYOUTS:
        raise NotImplementedError(f"dataset type {layout!r}: only the DTU / IDR folder layout of the reference's configs is built")
    kw = dict(train_cameras=False, data_dir=args.data.data_dir, downscale=args.data.downscale,
              scale_radius=args.data.get("scale_radius", -1), cam_file=args.data.get("cam_file", None))
    kw.update(overwrite_cfgs)
    sets = [_LAYOUTS[layout](**kw)]
    if return_val:
        sets.append(_LAYOUTS[layout](**{**kw, "downscale": val_downscale}))
    return tuple(sets) if return_val else sets[0]
