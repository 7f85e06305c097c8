"""Whole training steps of the three stages as the reference runs them -- the oracle's op graph + torch autograd + torch
Adam -- for timing baselines and full-size parity checks.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  The tensors decide where it runs: CPU tensors give the `cpu_baseline` leg of
bench.py (the reference's CPU path restated), device tensors give "the reference's op graph as plain PyTorch-ROCm ops on the
same MI355X" (the denominator of the north-star's ">= 10x the reference PyTorch path").
  stage 1: 1st_State-Conditional_Scene/src/model/mipnerf360/model.py:491-514 (+ run.py:155 norm clipping)
  stage 2: 2nd_State_Conditional_Human-Object/src/model/mipnerf360/model.py:571-634 (+ run.py:185-186 norm clipping)
  stage 3: 3rd_Complete_HOSNeRF/src/model/mipnerf360/model.py:1501-1658 (+ run.py:188-189 norm clipping)
Every launcher hands `run.grad_max_norm` (0.001 in all three Backpack.gin files) to the Lightning Trainer as
`gradient_clip_val` with algorithm "norm": `clip_grad_norm_` over ALL parameters of the step's single optimiser, after
backward and before `optimizer.step` -- in stage 3 one norm over the background model and the human network together.
The optimiser of stages 2/3 is ONE Adam with a param group per parameter whose rate is `cfg.train.lr_<module>`
(core/train/optimizers/human_nerf/optimizer.py:19-60): canonical MLP, state embeddings (and, stage 3, the background model)
at the base rate, the other human modules at a tenth of it (configs/default.yaml).
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


GRAD_MAX_NORM = 0.001          # run.grad_max_norm of the three Backpack.gin files


def _human_groups(p: Dict[str, torch.Tensor], lr: float):
    """optimizer.py:24-40 + default.yaml: one group per parameter; `cnl_mlp` / `human_stateembeds` at lr, the rest at lr / 10."""
    return [{"params": [v], "lr": lr if ("cnl_mlp" in k or "human_stateembeds" in k) else lr / 10.0, "name": k} for k, v in p.items()]


def stage1_step(sd, batch, device="cpu", lr: float = 2e-3, train_frac: float = 0.5, transitions=(0.4,)) -> Callable[[], torch.Tensor]:
    p = _params(sd, device)
    b = _to(batch, device)
    opt = torch.optim.Adam(list(p.values()), lr=lr)

    def step():
        opt.zero_grad()
        rend, hist = ob.mipnerf360_forward(p, b, train_frac, True, 0.1, 1e6, transitions_times=list(transitions))
        loss, _ = ob.stage1_loss(rend[-1]["rgb"], b["target"], hist)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(p.values()), GRAD_MAX_NORM)
        opt.step()
        return loss.detach()

    return step


def stage2_step(sd, batch, device="cpu", lr: float = 6.667e-4, transitions=(0.4,), t_rand: Optional[torch.Tensor] = None,
                grad_max_norm: float = GRAD_MAX_NORM, dtype=torch.float32, params_out: Optional[dict] = None):
    p = _params(sd, device, dtype)
    b = _to(batch, device)
    if dtype != torch.float32:
        b = {k: (v.to(dtype) if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in b.items()}
    time = float(batch["time"])
    opt = torch.optim.Adam(_human_groups(p, lr), lr=lr, betas=(0.9, 0.999))
    tr = None if t_rand is None else t_rand.to(device)
    if params_out is not None:
        params_out.update(p)

    def step():
        opt.zero_grad()
        B = b["near"].shape[0]
        draws = tr if tr is not None else torch.rand(B, 128, device=b["near"].device, dtype=dtype)
        out = oh.human_forward(p, b, transitions_times=list(transitions), t_rand=draws, stage=2)
        loss, _ = ol.stage2_losses(out, b, time)
        loss.backward()
        if grad_max_norm > 0:
            torch.nn.utils.clip_grad_norm_([v for v in p.values() if v.grad is not None], grad_max_norm)
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


def stage3_step(bsd, hsd, batch, device="cpu", lr: float = 6.667e-5, transitions=(0.4,), grad_max_norm: float = GRAD_MAX_NORM,
                t_rand: Optional[torch.Tensor] = None, jitters=None, dtype=torch.float32, params_out: Optional[dict] = None):
    pb, ph = _params(bsd, device, dtype), _params(hsd, device, dtype)
    b = _to(batch, device)
    if dtype != torch.float32:
        b = {k: (v.to(dtype) if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in b.items()}
    time = float(batch["time"])
    # ONE Adam over both modules (optimizer.py:42-58): human groups + the background parameters at lr_bkgd (= lr)
    opt = torch.optim.Adam(_human_groups(ph, lr) + [{"params": [v], "lr": lr, "name": k} for k, v in pb.items()], lr=lr, betas=(0.9, 0.999))
    tr = None if t_rand is None else t_rand.to(device)
    if params_out is not None:
        params_out.update({"model." + k: v for k, v in pb.items()})
        params_out.update({"human." + k: v for k, v in ph.items()})

    def step():
        opt.zero_grad()
        B = b["near"].shape[0]
        draws = tr if tr is not None else torch.rand(B, 128, device=b["near"].device, dtype=dtype)
        out = stage3_render(pb, ph, b, transitions, t_rand=draws, jitters=jitters)
        loss, _ = ol.stage3_losses(out, b, time)
        loss.backward()
        if grad_max_norm > 0:         # ONE norm over every parameter that received a gradient (the proposal MLPs do not)
            torch.nn.utils.clip_grad_norm_([v for v in list(pb.values()) + list(ph.values()) if v.grad is not None], grad_max_norm)
        opt.step()
        return loss.detach()

    return step


# ------------------------------------------------------------------ trainers over CHANGING items (tests/test_gpu_convergence.py)
def stage1_trainer(sd, device="cpu", transitions=(0.4,), grad_max_norm: float = GRAD_MAX_NORM, near: float = 0.1, far: float = 1e6,
                   set_to_none: bool = True):
    """The stage-1 loop of the reference (M1:491-514 training_step, M1:536-569 Adam + per-step learning rate, run.py:155 norm
    clip) over a stream of batches: returns (params, step) with `step(batch, lr, train_frac, jitters) -> loss`.  `batch` holds
    device tensors (`rays_o`, `rays_d`, `viewdirs`, `radii`, `times`, `target`); `jitters` are the per-level per-ray draws."""
    p = _params(sd, device)
    opt = torch.optim.Adam(list(p.values()), lr=1.0)

    def step(batch, lr: float, train_frac: float, jitters=None):
        for g in opt.param_groups:                       # M1:551-567: optimizer_step rewrites every group's lr each step
            g["lr"] = lr
        opt.zero_grad(set_to_none=set_to_none)      # Lightning / torch 2.0.1 default: None -> Adam SKIPS parameters without a gradient
        rend, hist = ob.mipnerf360_forward(p, batch, train_frac, True, near, far, transitions_times=list(transitions), jitters=jitters)
        loss, _ = ob.stage1_loss(rend[-1]["rgb"], batch["target"], hist)
        loss.backward()
        if grad_max_norm > 0:
            torch.nn.utils.clip_grad_norm_(list(p.values()), grad_max_norm)
        opt.step()
        return loss.detach()

    return p, step


def stage2_trainer(sd, device="cpu", lr: float = 6.667e-4, transitions=(0.4,), grad_max_norm: float = GRAD_MAX_NORM):
    """The stage-2 loop (M2:571-634 + the per-parameter groups of optimizer.py:19-60) over a stream of items: returns
    (params, step) with `step(item, t_rand, lr_scale=1.0) -> loss`.  `item`: a stage-2 dataset item with device tensors."""
    p = _params(sd, device)
    groups = _human_groups(p, lr)
    base = [g["lr"] for g in groups]
    opt = torch.optim.Adam(groups, lr=lr, betas=(0.9, 0.999))

    def step(item, t_rand, lr_scale: float = 1.0):
        for g, b0 in zip(opt.param_groups, base):        # M2:606-634: every group's lr = its base * decay(step)
            g["lr"] = b0 * lr_scale
        opt.zero_grad()
        out = oh.human_forward(p, item, transitions_times=list(transitions), t_rand=t_rand, stage=2)
        loss, _ = ol.stage2_losses(out, item, float(item["time"]))
        loss.backward()
        if grad_max_norm > 0:
            torch.nn.utils.clip_grad_norm_([v for v in p.values() if v.grad is not None], grad_max_norm)
        opt.step()
        return loss.detach()

    return p, step


def stage3_trainer(bsd, hsd, device="cpu", lr: float = 6.667e-5, transitions=(0.4,), grad_max_norm: float = GRAD_MAX_NORM):
    """The stage-3 loop (M:1501-1658: both renderers + merged composite + get_loss, ONE Adam over both modules with the per-parameter
    groups of optimizer.py:19-60, ONE gradient norm) over a stream of items: returns (background params, human params, step) with
    `step(item, t_rand, jitters, lr_scale=1.0) -> loss`.  `jitters`: three [B,1] host tensors (H:364 draws on the host)."""
    pb, ph = _params(bsd, device), _params(hsd, device)
    groups = _human_groups(ph, lr) + [{"params": [v], "lr": lr, "name": k} for k, v in pb.items()]
    base = [g["lr"] for g in groups]
    opt = torch.optim.Adam(groups, lr=lr, betas=(0.9, 0.999))

    def step(item, t_rand, jitters, lr_scale: float = 1.0):
        for g, b0 in zip(opt.param_groups, base):        # M:1631-1656: every group's lr = its base * decay(step)
            g["lr"] = b0 * lr_scale
        opt.zero_grad()
        out = stage3_render(pb, ph, item, transitions, t_rand=t_rand, jitters=jitters)
        loss, _ = ol.stage3_losses(out, item, float(item["time"]))
        loss.backward()
        if grad_max_norm > 0:         # ONE norm over every parameter that received a gradient (the proposal MLPs do not)
            torch.nn.utils.clip_grad_norm_([v for v in list(pb.values()) + list(ph.values()) if v.grad is not None], grad_max_norm)
        opt.step()
        return loss.detach()

    return pb, ph, step
