"""Stage-3 HOSNeRF renderer: background mip-NeRF-360 (+) human-object branch, composited per ray.

Mirrors the renderer part of `LitMipNeRF360` of the reference's stage 3
(3rd_Complete_HOSNeRF/src/model/mipnerf360/model.py:1501-1629 training_step, and the same block repeated
in progress/test_metrics/allimgs_metrics/free_view at :746-814, :951-1018, :1155-1222, :1358-1425):
the two sub-modules keep the reference's attribute names `model` and `human`, so a stage-3 checkpoint's
`state_dict` keys (`model.mlps.*`, `human.*`) load unchanged (run.py:206-212 warm-start path).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from . import ops
from .human_nerf import Network, default_cfg
from .mipnerf360 import MipNeRF360


class HOSNeRF(nn.Module):
    def __init__(self, cfg=None, basedir: Optional[str] = None, near_bkg: float = 0.1, far_bkg: float = 1e6):
        super().__init__()
        cfg = default_cfg(basedir) if cfg is None else cfg
        self.cfg = cfg
        self.near_bkg, self.far_bkg = near_bkg, far_bkg
        self.model = MipNeRF360(cfg.basedir, opaque_background=True, render_levels=False)    # S3/configs/HOSNeRF/Backpack.gin
        self.human = Network(cfg, stage=3)

    def zero_grad(self, set_to_none: bool = False):
        """The gradients live in the two flat buffers (HIP weight-gradient kernels write there, not through autograd):
        zero them and keep every `p.grad` aliased -- nn.Module's default (set_to_none=True) would detach them."""
        self.model.store.zero_grad()
        self.human.store.zero_grad()

    def render(self, batch: Dict[str, torch.Tensor], randomized: bool = True, is_train: bool = True,
               jitters=None, t_rand=None, prologue=None, with_cycle: bool = True, static_cycle: bool = False) -> Dict[str, torch.Tensor]:
        """M:1507-1596 on one ray batch (keys of SURVEY Appendix B).  Returns the human dict + `rgb` [B,3],
        `idx_fg`, `total_order`, `human_weights_sorted` and the background `ray_history`."""
        batch_bkg = {"rays_o": batch["rays_o_bkg"], "rays_d": batch["rays_d_bkg"], "viewdirs": batch["viewdirs_bkg"],
                     "radii": batch["radii"], "times": batch["time"]}
        # the reference passes train_frac = 1.0 and randomized = True everywhere in stage 3 (M:1512-1516, M:720-723)
        _, hist = self.model(batch_bkg, 1.0, randomized, is_train, self.near_bkg, self.far_bkg, jitters=jitters)
        out = self.human(t_rand=t_rand, prologue=prologue, with_cycle=with_cycle, static_cycle=static_cycle, **batch)
        last = hist[-1]
        rgb, hw, idx_fg, order, zh = ops.merge_composite(
            last["tdist"], last["rgb"], last["density"], out["human_rgbsigma"], out["newsmpl_pts"], out["pts_mask"],
            batch["rays_o_bkg"], batch["rays_d_bkg"], batch["newsmpl_to_scale_world"])
        out.update(rgb=rgb, idx_fg=idx_fg, total_order=order, human_weights_sorted=hw, z_vals_human=zh, ray_history=hist)
        return out

    def render_bkg_only(self, batch_bkg: Dict[str, torch.Tensor], randomized: bool = False, is_train: bool = False,
                        jitters=None) -> torch.Tensor:
        """Rays that miss the human bounding box (M:818-836, :1434-1452): the last level's 32 samples through the
        NeRF-style composite `_raw2outputs` with an all-ones mask.  Returns rgb [B,3]."""
        _, hist = self.model(batch_bkg, 1.0, randomized, is_train, self.near_bkg, self.far_bkg, jitters=jitters)
        last = hist[-1]
        z = last["tdist"][..., :-1].contiguous()
        rgbsigma = torch.cat([last["rgb"], last["density"][..., None]], -1).contiguous()
        return ops.raw2outputs(rgbsigma, z, batch_bkg["rays_d"].contiguous(), None, None)[0]
