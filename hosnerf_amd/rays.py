"""Device-side per-frame ray set-up: the reference's `core/utils/camera_util.py` functions
(`get_rays_from_KRT`, `get_rays_from_KRT_bkg`, `rays_intersect_3d_bbox`, stage 3 lines 154-265) with the same names,
argument order and return order, but returning torch tensors on the MI355X instead of numpy arrays -- a dataset's
`__getitem__` can hand the renderer device tensors directly (same batch-dict keys) instead of shipping full-image ray
arrays through pinned memory every step."""
from __future__ import annotations

import ctypes
from typing import Tuple

import numpy as np
import torch

from ._lib import call, ptr


def _host3(x, shape):
    a = np.ascontiguousarray(np.asarray(x.detach().cpu() if isinstance(x, torch.Tensor) else x, dtype=np.float64))
    assert a.shape == shape, a.shape
    return a


def _fptr(a: np.ndarray):
    a32 = np.ascontiguousarray(a, dtype=np.float32)
    return a32, a32.ctypes.data_as(ctypes.c_void_p).value


def _camera(H, W, K, R, T, bkg: bool, device):
    Kinv = np.linalg.inv(_host3(K, (3, 3)))               # 3x3 inverse on the host, like the reference (C:178)
    k32, kp = _fptr(Kinv)
    r32, rp = _fptr(_host3(R, (3, 3)))
    t32, tp = _fptr(_host3(T, (3,)))
    dev = torch.device(device)
    n = H * W
    o = torch.empty(n, 3, device=dev)
    d = torch.empty(n, 3, device=dev)
    vd = torch.empty(n, 3, device=dev) if bkg else None
    rad = torch.empty(n, device=dev) if bkg else None
    call("hos_camera_rays", kp, rp, tp, H, W, ptr(o), ptr(d), ptr(vd), ptr(rad))
    return o, d, vd, rad, (k32, r32, t32)      # keep the host arrays alive until the launch has been issued


def get_rays_from_KRT(H: int, W: int, K, R, T, device="cuda") -> Tuple[torch.Tensor, torch.Tensor]:
    """C:154-183 -> rays_o, rays_d [H, W, 3]."""
    T = np.asarray(T.detach().cpu() if isinstance(T, torch.Tensor) else T).reshape(3)
    o, d, _, _, _keep = _camera(H, W, K, R, T, False, device)
    return o.view(H, W, 3), d.view(H, W, 3)


def get_rays_from_KRT_bkg(H: int, W: int, K, R, T, device="cuda"):
    """C:185-216 -> rays_o, rays_d, viewdirs [H, W, 3], radii [H, W, 1]."""
    T = np.asarray(T.detach().cpu() if isinstance(T, torch.Tensor) else T).reshape(3)
    o, d, vd, rad, _keep = _camera(H, W, K, R, T, True, device)
    return o.view(H, W, 3), d.view(H, W, 3), vd.view(H, W, 3), rad.view(H, W, 1)


def rays_intersect_3d_bbox(bounds, ray_o: torch.Tensor, ray_d: torch.Tensor):
    """C:219-265 -> near [N_valid], far [N_valid], mask_at_box [N] (bool).  `bounds`: dict with min_xyz / max_xyz or a
    [2,3] array.  Like the reference, tiny components of `ray_d` are clamped to 1e-5 in place."""
    if isinstance(bounds, dict):
        bounds = np.stack([np.asarray(bounds["min_xyz"]), np.asarray(bounds["max_xyz"])], axis=0)
    b32, bp = _fptr(np.asarray(bounds, dtype=np.float64).reshape(6))
    assert ray_o.is_cuda and ray_d.is_cuda and ray_o.is_contiguous() and ray_d.is_contiguous()
    n = ray_o.shape[0]
    near = torch.empty(n, device=ray_o.device)
    far = torch.empty(n, device=ray_o.device)
    mask = torch.empty(n, dtype=torch.uint8, device=ray_o.device)
    call("hos_rays_aabb", ptr(ray_o), ptr(ray_d), n, bp, ptr(near), ptr(far), mask.data_ptr())
    m = mask.bool()
    return near[m], far[m], m


# ------------------------------------------------------------------------------------------ training item: patches
def get_patch_ray_indices(N_patch: int, ray_mask: torch.Tensor, subject_mask: torch.Tensor, bbox_mask: torch.Tensor,
                          patch_size: int, H: int, W: int, sample_subject_ratio: float, rng=np.random, cut_by_box: bool = False):
    """`Dataset.get_patch_ray_indices` (core/data/human_nerf/train.py:225-332) with the masks resident on the device.

    The random decisions are the reference's, drawn from the same numpy stream in the same order (`rng.rand(1)`, then
    `rng.choice(n_candidates, size=[1], replace=False)` -- n_candidates is the only thing read back from the device, twice
    per item); everything that touches pixels stays on the device.  Returns (select_inds [N*P*P] int64 into the
    box-compacted ray arrays, pixel indices [N*P*P] into the frame, patch_masks [N,P,P] bool, patch_div_indices [N+1]).
    Quirks kept: a patch is NOT intersected with the box (T:323 "to keep the patch size"), so pixels outside it map to
    the previous box ray through `cumsum(ray_mask) - 1`, and -1 wraps to the last ray like a numpy index."""
    dev = ray_mask.device
    subject = subject_mask.reshape(-1).to(dev)
    excl = bbox_mask.reshape(-1).to(dev) & ~subject                                   # T:238-241
    cand = (torch.nonzero(subject).reshape(-1), torch.nonzero(excl).reshape(-1))      # np.where order = row major
    masked_indices = torch.cumsum(ray_mask.reshape(-1).to(torch.int64), 0) - 1         # T:327
    P = int(patch_size)
    dy, dx = torch.meshgrid(torch.arange(P, device=dev), torch.arange(P, device=dev), indexing="ij")
    pix = []
    for _ in range(N_patch):
        which = 0 if rng.rand(1)[0] < sample_subject_ratio else 1                      # T:256-259
        n = int(cand[which].shape[0])
        centre = cand[which][int(rng.choice(n, size=[1], replace=False)[0])]           # T:297-301
        cy, cx = centre // W, centre % W
        x_min = torch.clamp(cx - P // 2, 0, W - P)                                     # T:304-311
        y_min = torch.clamp(cy - P // 2, 0, H - P)
        pix.append(((y_min + dy) * W + (x_min + dx)).reshape(-1))
    pix = torch.cat(pix, 0)
    if cut_by_box:
        # STAGE 2 (2nd_State_Conditional_Human-Object/core/data/human_nerf/train.py:321-332): the patch rectangle is intersected
        # with the rays that hit the subject's box -- the selection is ragged (its length is the one extra value read back per
        # item), `patch_masks` marks the pixels that kept their ray and `patch_div_indices` delimits the patches
        keep = ray_mask.reshape(-1).to(dev).bool()[pix]
        patch_masks = keep.view(N_patch, P, P)
        per = patch_masks.view(N_patch, -1).sum(1).cpu()
        div = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(per, 0)])
        pix_kept = pix[keep]
        return masked_indices[pix_kept], pix, patch_masks, div
    sel = masked_indices[pix]
    sel = torch.where(sel < 0, sel + masked_indices[-1] + 1, sel)                      # numpy's negative index
    patch_masks = torch.ones(N_patch, P, P, dtype=torch.bool, device=dev)              # T:323, :330
    div = torch.arange(N_patch + 1, dtype=torch.int64) * (P * P)
    return sel, pix, patch_masks, div


_PATCH_KEYS = ("near", "far", "rays_o_bkg", "rays_d_bkg", "viewdirs_bkg", "radii", "ray_img", "ray_grid")


def sample_patch_rays(item: dict, img: torch.Tensor, subject_mask: torch.Tensor, N_patches: int, patch_size: int,
                      sample_subject_ratio: float, rng=np.random, cut_by_box: bool = False) -> dict:
    """`Dataset.sample_patch_rays` (T:410-436): pick the patches and gather every per-ray array of the item
    (`rays` [2,n,3] and the `_PATCH_KEYS` present, all box-compacted as `eval.frame_rays` returns them).  Returns the
    item with those arrays replaced by the selected rays plus `target_patches` [N,P,P,3], `patch_masks`,
    `patch_div_indices` and `target_rgbs` (= the gathered `ray_img`).  `cut_by_box=True` is the stage-2 dataset's form
    (S2 train.py:405-455): only the patch pixels whose rays hit the box are selected, `patch_masks` has holes."""
    H, W = int(item["img_height"]), int(item["img_width"])
    rm = item["ray_mask"]
    sel, pix, masks, div = get_patch_ray_indices(N_patches, rm, subject_mask, rm.view(H, W), patch_size, H, W,
                                                 sample_subject_ratio, rng, cut_by_box=cut_by_box)
    out = dict(item)
    out["rays"] = item["rays"].index_select(1, sel)
    for k in _PATCH_KEYS:
        if k in item:
            out[k] = item[k].index_select(0, sel)
    if "ray_img" in out:
        out["target_rgbs"] = out["ray_img"]
    out["target_patches"] = img.reshape(-1, img.shape[-1]).index_select(0, pix).view(N_patches, patch_size, patch_size, -1)
    out["patch_masks"] = masks
    out["patch_div_indices"] = div
    return out
