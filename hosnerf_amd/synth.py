"""Deterministic synthetic weights and ray batches for the BASELINE.json configurations.

There is no dataset and no checkpoint in the build/bench environment, so every measurement runs
on random-init weights of the reference architecture and seeded synthetic rays of the reference
batch-dict shapes.  numpy RandomState is used (not torch RNG) so
the same seed gives bit-identical tensors on every torch build -- the golden fixtures under
tests/golden depend on that.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch

# ------------------------------------------------------------------ background weights


def _linear(rs: np.random.RandomState, n_out: int, n_in: int, kaiming: bool = True):
    """nn.Linear default bias init + kaiming_uniform_(a=0) weight (M:174-209)."""
    wb = math.sqrt(6.0 / n_in) if kaiming else 1.0 / math.sqrt(n_in)
    bb = 1.0 / math.sqrt(n_in)
    w = rs.uniform(-wb, wb, size=(n_out, n_in)).astype(np.float32)
    b = rs.uniform(-bb, bb, size=(n_out,)).astype(np.float32)
    return torch.from_numpy(w), torch.from_numpy(b)


def mlp_state_dict(rs, prefix: str, depth: int, width: int, n_states: int, rgb: bool,
                   pos_size: int = 568, skip: int = 4) -> Dict[str, torch.Tensor]:
    sd = {}
    fan = pos_size
    for i in range(depth):
        w, b = _linear(rs, width, fan)
        sd[f"{prefix}pts_linear.{i}.weight"], sd[f"{prefix}pts_linear.{i}.bias"] = w, b
        fan = width + pos_size if (i % skip == 0 and i > 0) else width
    sd[f"{prefix}density_layer.weight"], sd[f"{prefix}density_layer.bias"] = _linear(rs, 1, fan)
    if rgb:
        sd[f"{prefix}bottleneck_layer.weight"], sd[f"{prefix}bottleneck_layer.bias"] = _linear(rs, 256, fan)
        sd[f"{prefix}views_linear.0.weight"], sd[f"{prefix}views_linear.0.bias"] = _linear(rs, 128, 256 + 27)
        sd[f"{prefix}rgb_layer.weight"], sd[f"{prefix}rgb_layer.bias"] = _linear(rs, 3, 128)
    for k in range(n_states):
        sd[f"{prefix}bkgd_stateembeds.{k}"] = torch.from_numpy(rs.standard_normal(64).astype(np.float32))
    return sd


def background_state_dict(seed: int = 777, n_states: int = 2, prop_width: int = 256, nerf_width: int = 1024,
                          prop_depth: int = 4, nerf_depth: int = 8) -> Dict[str, torch.Tensor]:
    """state_dict of the reference `MipNeRF360` (keys `mlps.{0,1,2}.*`)."""
    rs = np.random.RandomState(seed)
    sd = {}
    sd.update(mlp_state_dict(rs, "mlps.0.", prop_depth, prop_width, n_states, rgb=False))
    sd.update(mlp_state_dict(rs, "mlps.1.", prop_depth, prop_width, n_states, rgb=False))
    sd.update(mlp_state_dict(rs, "mlps.2.", nerf_depth, nerf_width, n_states, rgb=True))
    return sd


# ------------------------------------------------------------------ stage-1 rays (config 2)


def stage1_batch(num_rays: int = 1024, seed: int = 777, time: float = 0.5) -> Dict[str, torch.Tensor]:
    """o ~ N(0,0.1^2), d unit, radii = 1e-3*U(0.5,2), viewdirs == rays_d."""
    rs = np.random.RandomState(seed)
    o = (rs.standard_normal((num_rays, 3)) * 0.1).astype(np.float32)
    d = rs.standard_normal((num_rays, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    radii = (1e-3 * rs.uniform(0.5, 2.0, size=(num_rays, 1))).astype(np.float32)
    target = rs.uniform(0, 1, size=(num_rays, 3)).astype(np.float32)
    t = torch.from_numpy
    return {
        "rays_o": t(o), "rays_d": t(d), "viewdirs": t(d.copy()), "radii": t(radii),
        "times": torch.full((num_rays,), float(time)), "target": t(target),
    }


def pinhole_batch(hw: int = 64, seed: int = 777, time: float = 0.5) -> Dict[str, torch.Tensor]:
    """hw x hw pin-hole camera, f = 1.2*hw, at z=+1 looking at the origin."""
    f = 1.2 * hw
    j, i = np.meshgrid(np.arange(hw, dtype=np.float32), np.arange(hw, dtype=np.float32), indexing="ij")
    dirs = np.stack([(i - hw / 2 + 0.5) / f, -(j - hw / 2 + 0.5) / f, -np.ones_like(i)], -1)
    d = dirs.reshape(-1, 3)
    dx = np.linalg.norm(dirs[:-1] - dirs[1:], axis=-1)
    dx = np.concatenate([dx, dx[-1:]], 0).reshape(-1, 1)
    radii = (dx * 2 / math.sqrt(12)).astype(np.float32)
    dn = d / np.linalg.norm(d, axis=-1, keepdims=True)
    o = np.broadcast_to(np.array([0, 0, 1], np.float32), d.shape).copy()
    rs = np.random.RandomState(seed)
    target = rs.uniform(0, 1, size=d.shape).astype(np.float32)
    t = torch.from_numpy
    return {
        "rays_o": t(o), "rays_d": t(dn.astype(np.float32)), "viewdirs": t(dn.astype(np.float32).copy()),
        "radii": t(radii), "times": torch.full((d.shape[0],), float(time)), "target": t(target),
    }
