"""Launcher registry of the reference (`utils/select_option.py::select_model` in each stage directory, called by
`run.py` with the gin-bound `model_name`): `state_mipnerf360` (stage 1), `state_humanobject` (stage 2), `hosnerf`
(stage 3).  The reference returns a Lightning module that owns the renderer(s) and the per-step logic; these light
wrappers expose the same attributes (`.model`, `.human`), `training_step(batch, batch_idx) -> loss` and
`configure_optimizers()`, so `run.py`'s Trainer loop (or a plain loop) can drive the HIP renderers unchanged.
They subclass `pytorch_lightning.LightningModule` when Lightning is installed, `nn.Module` otherwise (this image).

Datasets, LPIPS/SSIM evaluation and image dumps stay with the reference (out of scope)."""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import nn

try:  # pragma: no cover - Lightning is not in the build image
    import pytorch_lightning as pl  # type: ignore
    _Base = pl.LightningModule
except Exception:
    _Base = nn.Module

from .hosnerf import HOSNeRF
from .human_nerf import Network, default_cfg
from .mipnerf360 import MipNeRF360
from .train import (FusedAdam, FusedAdamOptimizer, GradClip, backward_human, finish_backward_human, human_lr_decay, human_lr_from_cfg, lr_ranges_by_name, stage1_loss,
                    stage1_lr, stage2_losses, stage3_losses)


class _LitFlat(_Base):
    """Shared plumbing of the three stage modules: the flat stores behind `self._flat_modules()` own the gradients, so
    `zero_grad` / `optimizer_zero_grad` (whatever `set_to_none` says) zero the flat buffers and keep every `p.grad` aliased
    to them -- nn.Module / Lightning defaults would drop the aliases and the HIP weight gradients would never reach Adam."""

    def _flat_modules(self):
        raise NotImplementedError

    def zero_grad(self, set_to_none: bool = False):
        for m in self._flat_modules():
            m.store.zero_grad()

    def optimizer_zero_grad(self, *args, **kwargs):                 # Lightning hook (epoch, batch_idx, optimizer[, idx])
        self.zero_grad()

    def _global_step(self) -> int:
        tr = getattr(self, "trainer", None) if _Base is not nn.Module else None
        return int(getattr(tr, "global_step", self._step))

    LR_BEFORE_STEP = False

    def optimizer_step(self, epoch=None, batch_idx=None, optimizer=None, *args, optimizer_closure=None, **kwargs):
        """Lightning hook: the step and the schedule, in the reference's order -- stage 1 writes lr(global_step) into the param
        groups and THEN steps (M1:541-569: the first update runs at the warm-up rate lr_init * lr_delay_mult), stages 2 / 3 step
        and then write the decayed rates for the next step (M2:606-634, M:1631-1658)."""
        closure = optimizer_closure if optimizer_closure is not None else (args[1] if len(args) > 1 and callable(args[1]) else None)
        step = self._global_step()          # the reference reads trainer.global_step BEFORE optimizer.step: the index of this step
        human = getattr(self, "human", None)
        if closure is None and human is not None and human.pending_volume_grad() is not None:
            # the loop called loss.backward() itself instead of this module's `backward` hook: finish the split backward here
            finish_backward_human(human, self._human_opt, getattr(self._human_opt, "group", None))
        if self.LR_BEFORE_STEP:
            self.apply_lr(optimizer, step)
            optimizer.step(closure=closure)
        else:
            optimizer.step(closure=closure)
            self.apply_lr(optimizer, step)
        self._step += 1                     # without a Trainer this module counts the optimiser steps itself
        if self.RANGE_POLL_EVERY > 0 and self._step % self.RANGE_POLL_EVERY == 0:
            self.poll_range_guard()

    RANGE_POLL_EVERY = 200

    def poll_range_guard(self) -> bool:
        """Every RANGE_POLL_EVERY steps: did the optimiser kernel skip steps because a hidden activation left the fp16 hi/lo range?
        If so the flat modules continue in exact fp32 MFMA (train.check_range) and the loop says so."""
        mods = self._flat_modules()
        dev = mods[0].flat_param.device
        if dev.type != "cuda" or torch.cuda.is_current_stream_capturing():
            return False
        from .train import check_range
        if check_range(mods, dev):
            import warnings
            warnings.warn(f"{type(self).__name__}: optimizer steps were skipped by the fp16 range guard; continuing in exact fp32 MFMA")
            return True
        return False

    def apply_lr(self, optimizer, step: int):
        raise NotImplementedError

    _human_opt: Optional[FusedAdam] = None

    def backward(self, loss, *args, **kwargs):
        """Lightning's `backward` hook (a plain loop calls it instead of `loss.backward()`).  Modules that own the human network
        run the data-parallel form: its volume decoder -- 253 of its 259 MB of parameters, fed by a learned constant, no ray
        enters it -- is reduced at its 3.5 MB OUTPUT gradient and backpropagated on the sum (`train.backward_human`), the other
        4.3 MB are all-reduced as one span; the optimiser step that follows skips the whole-buffer exchange."""
        human = getattr(self, "human", None)
        if human is not None and human.split_decoder_backward:
            backward_human(human, loss, self._human_opt, getattr(self._human_opt, "group", None))
        else:
            loss.backward()


class LitMipNeRF360(_LitFlat):
    """Stage 1 (S1/src/model/mipnerf360/model.py:464-563): `self.model = MipNeRF360(basedir)`; one step =
    forward + Charbonnier/interlevel/distortion losses.  `configure_optimizers` returns the flat fused Adam behind a
    `torch.optim.Optimizer` face (zero_grad-safe, see train.FusedAdamOptimizer); `fused_optimizer()` the bare object."""

    LR_BEFORE_STEP = True

    def __init__(self, basedir, lr_init: float = 2.0e-3, lr_final: float = 2.0e-5, lr_delay_steps: int = 512,
                 lr_delay_mult: float = 0.01, max_steps: int = 500000, grad_max_norm: float = 0.001,
                 near: float = 0.1, far: float = 1e6):
        super().__init__()
        self.lr_init, self.lr_final, self.lr_delay_steps, self.lr_delay_mult = lr_init, lr_final, lr_delay_steps, lr_delay_mult
        self.max_steps, self.grad_max_norm, self.near, self.far = max_steps, grad_max_norm, near, far
        self.model = MipNeRF360(basedir, opaque_background=True)
        self._step = 0

    def _flat_modules(self):
        return [self.model]

    def training_step(self, batch: Dict[str, torch.Tensor], batch_idx: int = 0) -> torch.Tensor:
        step = self._global_step()
        rend, hist = self.model(batch, step / self.max_steps, True, True, self.near, self.far)
        loss, _ = stage1_loss(rend[-1]["rgb"], batch["target"], hist)
        return loss

    def learning_rate(self, step: int) -> float:
        return stage1_lr(step, self.max_steps, self.lr_init, self.lr_final, self.lr_delay_steps, self.lr_delay_mult)

    def apply_lr(self, optimizer, step: int):
        for g in optimizer.param_groups:
            g["lr"] = self.learning_rate(step)

    def configure_optimizers(self):
        # norm clipping (`gradient_clip_val=grad_max_norm`, S1/run.py:155) is applied by the trainer in the reference;
        # here it is part of the fused step
        return FusedAdamOptimizer(self.model, lr=self.lr_init, max_grad_norm=self.grad_max_norm)

    def fused_optimizer(self) -> FusedAdam:
        return FusedAdam(self.model, lr=self.lr_init, max_grad_norm=self.grad_max_norm)


class LitHumanObject(_LitFlat):
    """Stage 2 (2nd_State_Conditional_Human-Object/src/model/mipnerf360/model.py:447-634): `self.human = Network(cfg)` with
    the in-network composite; `training_step` = network forward + `get_loss` (0.2 MSE on the unpacked patches + 0.01 flow +
    0.01 cycle; the LPIPS term needs the VGG weights and stays with the reference), `optimizer_step` = Adam + the
    0.1 ** (step / 500k) decay of every group's base learning rate."""

    LR = 6.667e-4           # configs/default.yaml train.lr (cnl_mlp, human_stateembeds); the other modules train at LR / 10

    def __init__(self, basedir, cfg=None, grad_max_norm: float = 0.0):
        """`grad_max_norm`: `run.grad_max_norm` (0.001 in configs/human-object/Backpack.gin), which 2nd_.../run.py:185-186 hands
        to the Trainer as `gradient_clip_val` with algorithm "norm"; here it is part of the fused optimiser step."""
        super().__init__()
        self.cfg = default_cfg(basedir) if cfg is None else cfg
        self.grad_max_norm = float(grad_max_norm)
        self.human = Network(self.cfg, stage=2)
        self._step = 0

    def _flat_modules(self):
        return [self.human]

    def forward(self, **batch):
        return self.human(**batch)

    def _base_lr(self) -> float:
        return human_lr_from_cfg(self.cfg, self.LR)[0]

    def training_step(self, batch: Dict[str, torch.Tensor], batch_idx: int = 0) -> torch.Tensor:
        """M2:571-605.  `batch` is the dataset item (leading DataLoader dimension already stripped) after
        `train.prepare_patch_targets` + `train.batch_to_device`."""
        batch = dict(batch)
        batch["iter_val"] = torch.full((1,), float(self._global_step()))          # M2:576
        self.human.split_decoder_backward = self._human_opt is not None and torch.is_grad_enabled()
        out = self.human(static_cycle=True, **batch)
        loss, _ = stage2_losses(out, batch, lpips=getattr(self, "lpips", None))
        return loss

    def apply_lr(self, optimizer, step: int):
        for g in optimizer.param_groups:
            g["lr"] = self._base_lr() * human_lr_decay(step, int(getattr(getattr(self.cfg, "train", None), "lrate_decay", 500)))

    def configure_optimizers(self):
        return FusedAdamOptimizer(self.fused_optimizer())

    def fused_optimizer(self) -> FusedAdam:
        lr, lr_of = human_lr_from_cfg(self.cfg, self.LR)         # cfg.train.lr / lr_<module> (optimizer.py:19-60), shipped defaults otherwise
        self._human_opt = FusedAdam(self.human, lr=lr, lr_ranges=lr_ranges_by_name(self.human, lr_of, lr), max_grad_norm=self.grad_max_norm)
        return self._human_opt


class LitHOSNeRF(_LitFlat):
    """Stage 3 (S3/src/model/mipnerf360/model.py:1501-1656): both renderers + the merged composite; one step =
    render + 0.2 MSE + 0.01 flow + 0.01 cycle, and + 1.0 LPIPS once `self.lpips = hosnerf_amd.lpips.LPIPS.from_files(vgg16.pth,
    third_parties/lpips/weights/v0.1/vgg.pth)` has been set (the ImageNet VGG-16 weights are torchvision's download: not shipped)."""

    LR = 6.667e-5           # configs/default.yaml train.lr_bkgd / lr_cnl_mlp; the other human modules train at LR / 10

    def __init__(self, basedir, cfg=None, grad_max_norm: float = 0.0):
        """`grad_max_norm`: `run.grad_max_norm` (0.001 in configs/HOSNeRF/Backpack.gin) = the Trainer's `gradient_clip_val`
        (3rd_.../run.py:188-189): ONE norm over the parameters of both modules, which share the reference's single Adam."""
        super().__init__()
        self.cfg = default_cfg(basedir) if cfg is None else cfg
        self.grad_max_norm = float(grad_max_norm)
        net = HOSNeRF(self.cfg)
        # registered under the reference's attribute names so that `state_dict()` has its keys (`model.*`, `human.*`);
        # the composite renderer that owns the same two modules is kept as a plain attribute
        self.model = net.model
        self.human = net.human
        object.__setattr__(self, "net", net)
        self._step = 0

    def _flat_modules(self):
        return [self.model, self.human]

    def training_step(self, batch: Dict[str, torch.Tensor], batch_idx: int = 0) -> torch.Tensor:
        batch = dict(batch)
        batch["iter_val"] = torch.full((1,), float(self._global_step()))          # M:1506
        self.human.split_decoder_backward = self._human_opt is not None and torch.is_grad_enabled()
        out = self.net.render(batch, randomized=True, is_train=True, static_cycle=True)
        loss, _ = stage3_losses(out, batch, lpips=getattr(self, "lpips", None))
        return loss

    def _lr_bkgd(self) -> float:
        tr = getattr(self.cfg, "train", None) or {}
        return float(tr.get("lr_bkgd", self.LR) if isinstance(tr, dict) else getattr(tr, "lr_bkgd", self.LR))

    def apply_lr(self, optimizer, step: int):
        """M:1631-1656: every group's rate = its OWN base (cfg.train.lr_<name>; lr_bkgd for the background model) * decay.  One
        param group per flat module here: the group's `lr` is the module's base rate, the per-module ratios inside the human
        network are the `lr_ranges` multipliers its FusedAdam was built with (`fused_optimizers`)."""
        decay = human_lr_decay(step, int(getattr(getattr(self.cfg, "train", None), "lrate_decay", 500)))
        for g, base in zip(optimizer.param_groups, (self._lr_bkgd(), human_lr_from_cfg(self.cfg, self.LR)[0])):
            g["lr"] = base * decay

    def configure_optimizers(self):
        return FusedAdamOptimizer(list(self.fused_optimizers()))

    def fused_optimizers(self):
        lr_h, lr_of = human_lr_from_cfg(self.cfg, self.LR)
        clip = GradClip(self.grad_max_norm)                   # shared: one global norm over both flat gradients
        self._human_opt = FusedAdam(self.net.human, lr=lr_h, lr_ranges=lr_ranges_by_name(self.net.human, lr_of, lr_h), clip=clip)
        return (FusedAdam(self.net.model, lr=self._lr_bkgd(), clip=clip), self._human_opt)


_MODELS = {"state_mipnerf360": LitMipNeRF360, "state_humanobject": LitHumanObject, "hosnerf": LitHOSNeRF}


def select_model(model_name: str, basedir, **kwargs):
    """Same call as the reference's `select_model(model_name, basedir)`; unknown names raise (the reference `raise`s a
    str, i.e. a TypeError -- a ValueError with the known names is the useful equivalent)."""
    if model_name not in _MODELS:
        raise ValueError(f"Unknown model named {model_name}; known: {sorted(_MODELS)}")
    return _MODELS[model_name](basedir, **kwargs)


# ------------------------------------------------------------------------------------------ checkpoints
def lightning_checkpoint(lit: nn.Module, global_step: int = 0, epoch: int = 0, optimizer=None) -> Dict:
    """The part of a Lightning `.ckpt` the reference reads back (`pl_load(path)['state_dict']`, S3/run.py:206-212, and
    `trainer.fit(ckpt_path=...)`): the module's `state_dict` under the reference's key names, on the host, plus --
    like Lightning's `optimizer_states` -- the optimiser's moments and step count (flat layout, `FusedAdam.state_dict`)
    so that a resumed run continues with the same Adam state instead of restarting the bias correction at t = 1."""
    ck = {"state_dict": {k: v.detach().cpu().clone() for k, v in lit.state_dict().items()},
          "global_step": int(global_step), "epoch": int(epoch), "pytorch-lightning_version": "hosnerf_amd"}
    if optimizer is not None:
        opts = optimizer if isinstance(optimizer, (list, tuple)) else [optimizer]
        ck["optimizer_states"] = [o.state_dict() for o in opts]
    return ck


def save_checkpoint(lit: nn.Module, path: str, global_step: int = 0, epoch: int = 0, optimizer=None) -> None:
    torch.save(lightning_checkpoint(lit, global_step, epoch, optimizer), path)


def load_optimizer_states(path: str, optimizer) -> int:
    """Restore the optimiser state(s) saved by `save_checkpoint(..., optimizer=...)`; returns the checkpoint's global step."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    opts = optimizer if isinstance(optimizer, (list, tuple)) else [optimizer]
    for o, sd in zip(opts, ckpt.get("optimizer_states", [])):
        o.load_state_dict(sd)
    return int(ckpt.get("global_step", 0))


def load_checkpoint(lit: nn.Module, path: str, strict: bool = False):
    """`model.load_state_dict(pl_load(path)['state_dict'], strict=False)` (S3/run.py:206-212): a stage-2 checkpoint fills
    `human.*`, a stage-1 checkpoint `model.*`; called once per file for the stage-3 warm start.  Values are copied INTO
    the flat parameter store (the parameters are views of it), so optimiser state and captured graphs stay valid.
    Returns torch's (missing_keys, unexpected_keys)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    return lit.load_state_dict(sd, strict=strict)
