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
