"""Whole training steps of the three stages as the reference runs them -- the oracle's op graph + torch autograd + torch
Adam -- for timing baselines and full-size parity checks.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  The tensors decide where it runs: CPU tensors give the `cpu_baseline` leg of
bench.py (the reference's CPU path restated), device tensors give "the reference's op graph as plain PyTorch-ROCm ops on the
same MI355X" (the denominator of the north-star's ">= 10x the reference PyTorch path").
  stage 1: 1st_State-Conditional_Scene/src/model/mipnerf360/model.py:491-514 (+ run.py:155 norm clipping)
  stage 2: 2nd_State_Conditional_Human-Object/src/model/mipnerf360/model.py:571-634
  stage 3: 3rd_Complete_HOSNeRF/src/model/mipnerf360/model.py:1501-1658
The LPIPS term of stages 2/3 (third-party VGG weights, absent offline) is left out on both sides of every comparison.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch

from . import background as ob
from . import human as oh
from . import losses as ol


def _params(sd: Dict[str, torch.Tensor], device, dtype=torch.float32):
    return {k: v.to(device=device, dtype=dtype).clone().requires_grad_(True) for k, v in sd.items()}


def _to(batch: Dict, device):
    """`cpu_data_to_gpu` (M:1507): every tensor of the item moves, control scalars included."""
    return {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


def stage1_step(sd, batch, device="cpu", lr: float = 2e-3, train_frac: float = 0.5, transitions=(0.4,)) -> Callable[[], torch.Tensor]:
    p = _params(sd, device)
    b = _to(batch, device)
    opt = torch.optim.Adam(list(p.values()), lr=lr)

    def step():
        opt.zero_grad()
        rend, hist = ob.mipnerf360_forward(p, b, train_frac, True, 0.1, 1e6, transitions_times=list(transitions))
        loss, _ = ob.stage1_loss(rend[-1]["rgb"], b["target"], hist)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(p.values()), 0.001)
        opt.step()
        return loss.detach()

    return step


def stage2_step(sd, batch, device="cpu", lr: float = 6.667e-4, transitions=(0.4,), t_rand: Optional[torch.Tensor] = None):
    p = _params(sd, device)
    b = _to(batch, device)
    time = float(batch["time"])
    opt = torch.optim.Adam(list(p.values()), lr=lr)
    tr = None if t_rand is None else t_rand.to(device)

    def step():
        opt.zero_grad()
        B = b["near"].shape[0]
        draws = tr if tr is not None else torch.rand(B, 128, device=b["near"].device)
        out = oh.human_forward(p, b, transitions_times=list(transitions), t_rand=draws, stage=2)
        loss, _ = ol.stage2_losses(out, b, time)
        loss.backward()
        opt.step()
        return loss.detach()

    return step


def stage3_render(bsd, hsd, b, transitions=(0.4,), t_rand=None, jitters=None):
    """M:1507-1596 on the oracle: background history (only the NeRF level is rendered from), human samples, merged composite."""
    bb = {"rays_o": b["rays_o_bkg"], "rays_d": b["rays_d_bkg"], "viewdirs": b["viewdirs_bkg"], "radii": b["radii"], "times": b["time"]}
    _, hist = ob.mipnerf360_forward(bsd, bb, 1.0, True, 0.1, 1e6, transitions_times=list(transitions), jitters=jitters, render=False)
    human = oh.human_forward(hsd, b, transitions_times=list(transitions), t_rand=t_rand, stage=3)
    rgb, fg, order, hw, _ = oh.stage3_composite(hist[-1]["tdist"], hist[-1]["rgb"], hist[-1]["density"], human, bb["rays_o"], bb["rays_d"],
                                                b["newsmpl_to_scale_world"])
    out = dict(human, rgb=rgb, idx_fg=fg, human_weights_onlyfg=hw, total_order=order)
    return out


def stage3_step(bsd, hsd, batch, device="cpu", lr: float = 6.667e-5, transitions=(0.4,)):
    pb, ph = _params(bsd, device), _params(hsd, device)
    b = _to(batch, device)
    time = float(batch["time"])
    opt = torch.optim.Adam(list(pb.values()) + list(ph.values()), lr=lr)

    def step():
        opt.zero_grad()
        B = b["near"].shape[0]
        out = stage3_render(pb, ph, b, transitions, t_rand=torch.rand(B, 128, device=b["near"].device))
        loss, _ = ol.stage3_losses(out, b, time)
        loss.backward()
        opt.step()
        return loss.detach()

    return step
