"""Full-frame rendering loops of stage 3 (SURVEY 8(f).1 + 8(f).3).

The reference renders a frame in four near-identical methods of `LitMipNeRF360` -- `progress` (M:680-880), `test_metrics`
(M:884-1085), `allimgs_metrics` (M:1089-1289) and `free_view` (M:1293-1494), M = 3rd_Complete_HOSNeRF/src/model/mipnerf360/model.py:
    for each `chunk_bkg` = 8192 rays that hit the human bounding box:  background model + human network + z-merge composite
    for each `chunk_bkg` rays that miss it:                            background model + 32-sample `_raw2outputs`
    rendered[ray_mask] = rgb; rendered[ray_mask_bkg] = bkg_rgbs; PSNR against the ground-truth pixels.
The per-frame inputs come from a dataset `__getitem__` that builds full-image rays in numpy
(3rd_Complete_HOSNeRF/core/data/human_nerf/freeview.py:199-337).

Here the same loop is one function, `render_frame`, with three differences in HOW (not WHAT) it computes:
  * rays, radii and the box test are produced on the device by `frame_rays` (hos_rays.hip) under the dataset's batch keys;
  * the ray-independent prologue of the human network (pose refinement, motion bases, the 253 MB motion-weight volume
    decoder) runs once per frame instead of once per chunk (`Network.frame_prologue`);
  * with a process group the two ray lists are split into contiguous per-rank ranges and the colours are all-gathered
    (`train.shard_frame` / `train.gather_frame`, the role of `alter_gather_cat`, S1/src/model/interface.py:30-39).
"""
from __future__ import annotations

import contextlib
import math
from typing import Dict, Optional

import numpy as np
import torch

from . import rays as rays_mod
from .train import gather_frame, shard_frame

# per-frame (ray independent) keys of the human network's batch (M:1337-1352)
FRAME_KEYS = ("bgcolor", "dst_Rs", "dst_Ts", "cnl_gtfms", "canonical_joints", "motion_weights_priors", "cnl_bbox_min_xyz",
              "cnl_bbox_max_xyz", "cnl_bbox_scale_xyz", "dst_posevec", "iter_val", "time", "is_train", "newsmpl_to_scale_world")


@contextlib.contextmanager
def evaluating(hos):
    """`test_begin` / `test_end` (M:670-678): eval mode, no depth jitter in the human branch, no autograd."""
    was_training = hos.training
    perturb = hos.cfg.perturb
    hos.eval()
    hos.cfg.perturb = 0.0
    try:
        with torch.no_grad():
            yield hos
    finally:
        hos.cfg.perturb = perturb
        hos.train(was_training)


def frame_rays(H: int, W: int, K, E, dst_bbox, E_colmap, device="cuda") -> Dict[str, torch.Tensor]:
    """The ray part of `FreeviewDataset.__getitem__` (freeview.py:239-283) on the device.  `E` is the SMPL-space camera
    (after `apply_global_tfm_to_camera`), `E_colmap` the background-world camera; both [4,4] or [3,4] host arrays."""
    E = np.asarray(E, dtype=np.float64)
    Ec = np.asarray(E_colmap, dtype=np.float64)
    o, d = rays_mod.get_rays_from_KRT(H, W, K, E[:3, :3], E[:3, 3], device=device)
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    near, far, ray_mask = rays_mod.rays_intersect_3d_bbox(dst_bbox, o, d)
    ob, db, vb, rad = rays_mod.get_rays_from_KRT_bkg(H, W, K, Ec[:3, :3], Ec[:3, 3], device=device)
    ob, db, vb, rad = ob.reshape(-1, 3), db.reshape(-1, 3), vb.reshape(-1, 3), rad.reshape(-1, 1)
    miss = ~ray_mask
    return {
        "img_width": W, "img_height": H, "ray_mask": ray_mask, "ray_mask_bkg": miss,
        "rays": torch.stack([o[ray_mask], d[ray_mask]], 0), "near": near[:, None], "far": far[:, None],
        "rays_o_bkg": ob[ray_mask], "rays_d_bkg": db[ray_mask], "viewdirs_bkg": vb[ray_mask], "radii": rad[ray_mask],
        "rays_o_bkg_only": ob[miss], "rays_d_bkg_only": db[miss], "viewdirs_bkg_only": vb[miss], "radii_bkg_only": rad[miss],
    }


def _local(tensors: Dict[str, torch.Tensor], n: int, group) -> Dict[str, torch.Tensor]:
    """This rank's contiguous share of `n` rays (padded by repeating the last ray, like S1/src/data/interface.py:152-166)."""
    if group is None or n == 0:
        return tensors
    import torch.distributed as dist
    idx, _ = shard_frame(n, dist.get_rank(group), dist.get_world_size(group))
    idx = idx.to(next(iter(tensors.values())).device)
    return {k: v.index_select(1 if k == "rays" else 0, idx) for k, v in tensors.items()}


def render_frame(hos, frame: Dict, chunk_bkg: int = 8192, randomized: bool = False, group=None,
                 cache_prologue: bool = True) -> torch.Tensor:
    """One frame of `free_view` / `test_metrics` / `progress` (M:1320-1458).  `frame` carries the keys of the reference's
    evaluation batch (freeview.py:284-335).  Returns `rendered` [H*W, 3] on the device (every rank holds the whole frame
    when `group` is given).  `randomized` is False in free_view/test_metrics and True in progress (M:720-723)."""
    H, W = int(frame["img_height"]), int(frame["img_width"])
    dev = frame["rays_o_bkg"].device
    per_frame = {k: frame[k] for k in FRAME_KEYS if k in frame}
    per_frame["is_train"] = False
    with evaluating(hos):
        # ---- rays through the human box: both branches + merge (M:1322-1432)
        n_fg = frame["rays_o_bkg"].shape[0]
        fg = _local({k: frame[k] for k in ("rays", "near", "far", "rays_o_bkg", "rays_d_bkg", "viewdirs_bkg", "radii")}, n_fg, group)
        pro = hos.human.frame_prologue(**per_frame) if (cache_prologue and n_fg > 0) else None
        parts = []
        for i in range(0, fg["near"].shape[0], chunk_bkg):
            sl = slice(i, i + chunk_bkg)
            b = dict(per_frame)
            b.update({k: (v[:, sl] if k == "rays" else v[sl]).contiguous() for k, v in fg.items()})
            parts.append(hos.render(b, randomized=randomized, is_train=False, prologue=pro, with_cycle=False)["rgb"])
        rgb = torch.cat(parts, 0) if parts else torch.zeros(0, 3, device=dev)
        # ---- rays that miss it: background only (M:1434-1452)
        n_bg = frame["rays_o_bkg_only"].shape[0]
        bg_rays = _local({"rays_o": frame["rays_o_bkg_only"], "rays_d": frame["rays_d_bkg_only"],
                          "viewdirs": frame["viewdirs_bkg_only"], "radii": frame["radii_bkg_only"]}, n_bg, group)
        parts = []
        for i in range(0, bg_rays["radii"].shape[0], chunk_bkg):
            bb = {k: v[i:i + chunk_bkg].contiguous() for k, v in bg_rays.items()}
            bb["times"] = frame["time"]
            parts.append(hos.render_bkg_only(bb, randomized=randomized, is_train=False))
        bkg_rgbs = torch.cat(parts, 0) if parts else torch.zeros(0, 3, device=dev)
        if group is not None:
            rgb = gather_frame(rgb, n_fg, group) if n_fg else rgb
            bkg_rgbs = gather_frame(bkg_rgbs, n_bg, group) if n_bg else bkg_rgbs
    bg = torch.as_tensor(frame.get("bgcolor", (0.0, 0.0, 0.0)), dtype=torch.float32, device=dev).reshape(3) / 255.0
    rendered = bg.expand(H * W, 3).clone()                                        # M:1311-1313
    rendered[frame["ray_mask"]] = rgb                                             # M:1456-1459
    rendered[frame["ray_mask_bkg"]] = bkg_rgbs
    return rendered


def psnr_metric(img_pred: torch.Tensor, img_gt: torch.Tensor) -> float:
    """M:101-112: -10 log10(mean squared error) over the whole frame, images in [0, 1]."""
    mse = torch.mean((img_pred.double() - img_gt.double()) ** 2).item()
    return -10.0 * math.log(mse) / math.log(10.0)


def truth_frame(frame: Dict) -> torch.Tensor:
    """`truth` of M:1314-1316, :1457-1460 from `target_rgbs` / `target_rgbs_bkg`."""
    H, W = int(frame["img_height"]), int(frame["img_width"])
    dev = frame["rays_o_bkg"].device
    bg = torch.as_tensor(frame.get("bgcolor", (0.0, 0.0, 0.0)), dtype=torch.float32, device=dev).reshape(3) / 255.0
    truth = bg.expand(H * W, 3).clone()
    truth[frame["ray_mask"]] = frame["target_rgbs"].to(dev).float()
    truth[frame["ray_mask_bkg"]] = frame["target_rgbs_bkg"].to(dev).float()
    return truth


def to_8b_image(image: torch.Tensor) -> torch.Tensor:
    """core/utils/image_util.py:28-29."""
    return (255.0 * image.clamp(0.0, 1.0)).to(torch.uint8)
