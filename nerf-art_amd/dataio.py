"""Scene dataset + camera loading: the data formats on the input side of the renderer.

Mirrors the reference's `dataio.get_data` / `dataio/DTU.py:SceneDataset` (IDR-style scene folder: `images/`, `matte/`,
`cameras.npz` holding `world_mat_i`, `scale_mat_i`) and the helpers they use (`utils/io_util.py:19-56` glob_imgs /
load_rgb / load_mask, `utils/rend_util.py:8-25` load_K_Rt_from_P) - same names, arguments, attributes and return
values, so `render.py` / `train.py` style callers run unchanged.  The reference leans on cv2, imageio and skimage for
three things; none of them is in this image, so they are restated on numpy / scipy / PIL / torch:

* `cv2.decomposeProjectionMatrix` -> RQ decomposition with cv2's conventions (positive diagonal of K, camera centre =
  null vector of P).  PARITY UNPINNED against cv2 itself (absent); pinned by properties on the reference's own
  `data/fangzhou_nature/cameras.npz` cameras (tests/test_dataio.py: K [R | -R c] reproduces P, R orthonormal).
* `imageio.imread` -> PIL; `skimage.img_as_float32` -> / 255.
* `skimage.transform.rescale(img, 1 / downscale, anti_aliasing=False)` (order 1, half-pixel centres) ->
  `F.interpolate(mode="bilinear", align_corners=False, antialias=False)`, the same sampling rule (for an integer
  downscale every output pixel is the mean of a 2 x 2 neighbourhood of input pixels in both).  UNPINNED as well.
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


def load_K_Rt_from_P(P):
    """P [3, 4] = K [R | t] -> (intrinsics [4, 4] float64 with K / K[2,2], pose [4, 4] float32 = camera-to-world)
    (rend_util.py:8-25).  K upper triangular with a positive diagonal, R = K^-1 P[:, :3], centre c = -P[:, :3]^-1 P[:, 3]."""
    from scipy.linalg import rq
    P = np.asarray(P, dtype=np.float64)
    M = P[:3, :3]
    K, R = rq(M)
    D = np.diag(np.where(np.diag(K) < 0, -1.0, 1.0))
    K, R = K @ D, D @ R                                   # (K D)(D R) = K R: flips columns of K / rows of R
    c = -np.linalg.solve(M, P[:3, 3])
    intrinsics = np.eye(4)
    intrinsics[:3, :3] = K / K[2, 2]
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R.transpose()
    pose[:3, 3] = c
    return intrinsics, pose


class SceneDataset(torch.utils.data.Dataset):
    """dataio/DTU.py:11-155: one item per image - (idx, {"object_mask" [H W] bool, "intrinsics" [4,4], "c2w" [4,4]},
    {"rgb" [H W, 3]}); cameras are scaled so that the farthest sits at scale_radius / 1.1 (DTU.py:68-71)."""

    def __init__(self, train_cameras, data_dir, downscale=1., cam_file=None, scale_radius=-1):
        assert os.path.exists(data_dir), f"Data directory {data_dir} is empty"
        self.instance_dir = data_dir
        self.train_cameras = train_cameras
        image_paths = sorted(glob_imgs("{0}/images".format(self.instance_dir)))
        mask_paths = sorted(glob_imgs("{0}/matte".format(self.instance_dir)))       # only the NeuS + mask setting uses them
        self.n_images = len(image_paths)
        self.downscale = downscale
        _, self.H, self.W = load_rgb(image_paths[0], downscale).shape
        self.cam_file = "{0}/cameras.npz".format(self.instance_dir)
        if cam_file is not None:
            self.cam_file = "{0}/{1}".format(self.instance_dir, cam_file)
        camera_dict = np.load(self.cam_file)
        scale_mats = [camera_dict["scale_mat_%d" % idx].astype(np.float32) for idx in range(self.n_images)]
        world_mats = [camera_dict["world_mat_%d" % idx].astype(np.float32) for idx in range(self.n_images)]
        self.intrinsics_all, self.c2w_all, cam_center_norms = [], [], []
        for scale_mat, world_mat in zip(scale_mats, world_mats):
            intrinsics, pose = load_K_Rt_from_P((world_mat @ scale_mat)[:3, :4])
            cam_center_norms.append(np.linalg.norm(pose[:3, 3]))
            intrinsics[0, 2] /= downscale
            intrinsics[1, 2] /= downscale
            intrinsics[0, 0] /= downscale
            intrinsics[1, 1] /= downscale                 # the skew is a ratio and is not scaled (DTU.py:63)
            self.intrinsics_all.append(torch.from_numpy(intrinsics).float())
            self.c2w_all.append(torch.from_numpy(pose).float())
        max_cam_norm = max(cam_center_norms)
        if scale_radius > 0:
            for i in range(len(self.c2w_all)):
                self.c2w_all[i][:3, 3] *= (scale_radius / max_cam_norm / 1.1)
        self.rgb_images = []
        for path in image_paths:
            rgb = load_rgb(path, downscale).reshape(3, -1).transpose(1, 0)
            self.rgb_images.append(torch.from_numpy(np.ascontiguousarray(rgb)).float())
        self.object_masks = []
        for path in mask_paths:
            self.object_masks.append(torch.from_numpy(load_mask(path, downscale).reshape(-1)).to(dtype=torch.bool))

    def __len__(self):
        return self.n_images

    def __getitem__(self, idx):
        sample = {"object_mask": self.object_masks[idx], "intrinsics": self.intrinsics_all[idx]}
        ground_truth = {"rgb": self.rgb_images[idx]}
        if not self.train_cameras:
            sample["c2w"] = self.c2w_all[idx]
        return idx, sample, ground_truth

    def collate_fn(self, batch_list):
        """list of (idx, dict, dict) -> (LongTensor, stacked dict, stacked dict)  (DTU.py:111-127)."""
        all_parsed = []
        for entry in zip(*batch_list):
            if type(entry[0]) is dict:
                all_parsed.append({k: torch.stack([obj[k] for obj in entry]) for k in entry[0].keys()})
            else:
                all_parsed.append(torch.LongTensor(entry))
        return tuple(all_parsed)

    def get_scale_mat(self):
        return np.load(self.cam_file)["scale_mat_0"]

    def get_gt_pose(self, scaled=True):
        """[n, 4, 4] camera-to-world without the scale_radius normalisation (DTU.py:132-147)."""
        camera_dict = np.load(self.cam_file)
        c2w_all = []
        for idx in range(self.n_images):
            P = camera_dict["world_mat_%d" % idx].astype(np.float32)
            if scaled:
                P = P @ camera_dict["scale_mat_%d" % idx].astype(np.float32)
            c2w_all.append(torch.from_numpy(load_K_Rt_from_P(P[:3, :4])[1]).float())
        return torch.stack(c2w_all, 0)


def get_data(args, return_val=False, val_downscale=4.0, **overwrite_cfgs):
    """dataio/__init__.py:1-26 for the 'DTU' layout every reference config uses."""
    dataset_type = args.data.get("type", "DTU")
    if dataset_type != "DTU":
        raise NotImplementedError(f"dataset type {dataset_type!r}: only the DTU / IDR folder layout of the reference's configs is built")
    cfgs = {"scale_radius": args.data.get("scale_radius", -1), "downscale": args.data.downscale, "data_dir": args.data.data_dir,
            "train_cameras": False, "cam_file": args.data.get("cam_file", None)}
    cfgs.update(overwrite_cfgs)
    dataset = SceneDataset(**cfgs)
    if return_val:
        cfgs["downscale"] = val_downscale
        return dataset, SceneDataset(**cfgs)
    return dataset
