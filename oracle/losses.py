"""Oracle (CPU, torch fp32) for the training losses of the human-object stages.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Abbreviations:
  M:  = 3rd_Complete_HOSNeRF/src/model/mipnerf360/model.py              (stage 3)
  M2: = 2nd_State_Conditional_Human-Object/src/model/mipnerf360/model.py (stage 2)
Pinned by tests/golden/losses.npz (values and gradients of the reference's own `get_loss`, LPIPS term switched off,
tests/golden/make_golden_losses.py).
"""
from __future__ import annotations

from typing import Dict

import torch


def img2mae(x, y, weights=None, M=None):
    """M:61-71 / M2:61-71."""
    if weights is None:
        if M is None:
            return torch.mean(torch.abs(x - y))
        return torch.sum(torch.abs(x - y) * M) / (torch.sum(M) + 1e-8) / x.shape[-1]
    if M is None:
        return torch.mean(torch.abs(x - y) * weights[..., None])
    return torch.sum(torch.abs(x - y) * weights[..., None] * M) / (torch.sum(M) + 1e-8) / x.shape[-1]


def flow_func(ray_grid, newsmpl_to_camera_prev, intrinsics_prev, weights, deform_pts_prev_final):
    """M:1680-1688 / M2:908-916: project the forward-warped previous-frame points with the previous camera and compare the
    induced flow with the dataset's optical flow, weighted by the composite weights and the flow validity mask."""
    hom = torch.cat([deform_pts_prev_final, torch.ones_like(deform_pts_prev_final[..., :1])], -1)
    cam = torch.einsum("ji,bni->bnj", newsmpl_to_camera_prev, hom)[..., :3]
    uvw = torch.einsum("ji,bni->bnj", intrinsics_prev, cam)
    uv = uvw[..., :-1] / uvw[..., -1:]
    grid = ray_grid.unsqueeze(1).repeat(1, uv.shape[1], 1)
    induced = uv - grid[..., :2]
    return img2mae(induced, grid[..., 2:4], weights, grid[..., -1].unsqueeze(-1))


def unpack_imgs(rgbs, patch_masks, bgcolor, targets, div_indices):
    """M:41-50 / M2:41-50: scatter the rendered rays back into their patches; pixels outside the ray mask hold bgcolor."""
    n = len(div_indices) - 1
    imgs = bgcolor.expand(targets.shape).clone()
    for i in range(n):
        imgs[i, patch_masks[i]] = rgbs[int(div_indices[i]):int(div_indices[i + 1])]
    return imgs


def stage3_losses(out: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor], time: float, w_mse=0.2, w_flow=0.01, w_cycle=0.01):
    """M:1690-1716 `get_loss` without the LPIPS term.  `out`: rgb [B,3], idx_fg [B] bool, human_weights_onlyfg [B_fg,S],
    deform_pts_prev_final [B,S,3], observe_pts, deform_pts_final.  Returns (total, {name: unweighted term})."""
    targets = batch["target_patches"]
    rgb = unpack_imgs(out["rgb"], batch["patch_masks"], batch["bgcolor"] / 255.0, targets, batch["patch_div_indices"])
    losses = {"mse": torch.mean((rgb - targets) ** 2)}
    if time > 0.005:
        fg = out["idx_fg"].bool()
        losses["flow"] = flow_func(batch["ray_grid"][fg], batch["newsmpl_to_camera_prev"], batch["intrinsics_prev"],
                                   out["human_weights_onlyfg"], out["deform_pts_prev_final"][fg])          # M:1703-1704
    else:
        losses["flow"] = torch.zeros(())
    dis = out["observe_pts"] - out["deform_pts_final"]
    losses["cycle"] = torch.mean(torch.sum(dis**2, 1) / 2.0)
    return w_mse * losses["mse"] + w_flow * losses["flow"] + w_cycle * losses["cycle"], losses


def stage2_losses(out: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor], time: float, w_mse=0.2, w_flow=0.01, w_cycle=0.01):
    """M2:918-944 `get_loss` without the LPIPS term: the flow term uses the network's own composite weights
    (`net_output['weights']`, all rays)."""
    targets = batch["target_patches"]
    rgb = unpack_imgs(out["rgb"], batch["patch_masks"], batch["bgcolor"] / 255.0, targets, batch["patch_div_indices"])
    losses = {"mse": torch.mean((rgb - targets) ** 2)}
    if time > 0.005:
        losses["flow"] = flow_func(batch["ray_grid"], batch["newsmpl_to_camera_prev"], batch["intrinsics_prev"],
                                   out["weights"], out["deform_pts_prev_final"])
    else:
        losses["flow"] = torch.zeros(())
    dis = out["observe_pts"] - out["deform_pts_final"]
    losses["cycle"] = torch.mean(torch.sum(dis**2, 1) / 2.0)
    return w_mse * losses["mse"] + w_flow * losses["flow"] + w_cycle * losses["cycle"], losses
