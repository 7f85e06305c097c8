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
from .train import FusedAdam, human_lr_ranges, stage1_loss, stage1_lr, stage3_losses


class LitMipNeRF360(_Base):
    """Stage 1 (S1/src/model/mipnerf360/model.py:464-563): `self.model = MipNeRF360(basedir)`; one step =
    forward + Charbonnier/interlevel/distortion losses.  `configure_optimizers` returns a torch Adam over the
    parameters (their `.grad`s are views of the flat gradient buffer); `fused_optimizer()` the one-launch variant."""

    def __init__(self, basedir, lr_init: float = 2.0e-3, lr_final: float = 2.0e-5, lr_delay_steps: int = 512,
                 lr_delay_mult: float = 0.01, max_steps: int = 500000, grad_max_norm: float = 0.001,
                 near: float = 0.1, far: float = 1e6):
        super().__init__()
        self.lr_init, self.lr_final, self.lr_delay_steps, self.lr_delay_mult = lr_init, lr_final, lr_delay_steps, lr_delay_mult
        self.max_steps, self.grad_max_norm, self.near, self.far = max_steps, grad_max_norm, near, far
        self.model = MipNeRF360(basedir, opaque_background=True)
        self._step = 0

    def training_step(self, batch: Dict[str, torch.Tensor], batch_idx: int = 0) -> torch.Tensor:
        step = getattr(getattr(self, "trainer", None), "global_step", self._step) if _Base is not nn.Module else self._step
        rend, hist = self.model(batch, step / self.max_steps, True, True, self.near, self.far)
        loss, _ = stage1_loss(rend[-1]["rgb"], batch["target"], hist)
        self._step += 1
        return loss

    def learning_rate(self, step: int) -> float:
        return stage1_lr(step, self.max_steps, self.lr_init, self.lr_final, self.lr_delay_steps, self.lr_delay_mult)

    def configure_optimizers(self):
        return torch.optim.Adam(self.parameters(), lr=self.lr_init, betas=(0.9, 0.999), eps=1e-8)

    def fused_optimizer(self) -> FusedAdam:
        return FusedAdam(self.model, lr=self.lr_init, max_grad_norm=self.grad_max_norm)


class LitHumanObject(_Base):
    """Stage 2: `self.human = Network(cfg)` (core/nets/human_nerf/network.py); the photometric / LPIPS losses of the
    reference's stage-2 trainer stay outside (they need its patch sampler); this wrapper owns renderer + optimiser."""

    def __init__(self, basedir, cfg=None):
        super().__init__()
        self.cfg = default_cfg(basedir) if cfg is None else cfg
        self.human = Network(self.cfg, stage=2)

    def forward(self, **batch):
        return self.human(**batch)

    def fused_optimizer(self) -> FusedAdam:
        return FusedAdam(self.human, lr=self.cfg.train.lr_cnl_mlp if hasattr(self.cfg, "train") else 6.667e-5,
                         lr_ranges=human_lr_ranges(self.human))


class LitHOSNeRF(_Base):
    """Stage 3 (S3/src/model/mipnerf360/model.py:1501-1656): both renderers + the merged composite; one step =
    render + 0.2 MSE + 0.01 flow + 0.01 cycle (the LPIPS term needs the VGG weights and stays with the reference)."""

    def __init__(self, basedir, cfg=None):
        super().__init__()
        self.cfg = default_cfg(basedir) if cfg is None else cfg
        net = HOSNeRF(self.cfg)
        # registered under the reference's attribute names so that `state_dict()` has its keys (`model.*`, `human.*`);
        # the composite renderer that owns the same two modules is kept as a plain attribute
        self.model = net.model
        self.human = net.human
        object.__setattr__(self, "net", net)

    def training_step(self, batch: Dict[str, torch.Tensor], batch_idx: int = 0) -> torch.Tensor:
        out = self.net.render(batch, randomized=True, is_train=True)
        loss, _ = stage3_losses(out, batch)
        return loss

    def fused_optimizers(self, lr: float = 6.667e-5):
        return (FusedAdam(self.net.model, lr=lr), FusedAdam(self.net.human, lr=lr, lr_ranges=human_lr_ranges(self.net.human)))


_MODELS = {"state_mipnerf360": LitMipNeRF360, "state_humanobject": LitHumanObject, "hosnerf": LitHOSNeRF}


def select_model(model_name: str, basedir, **kwargs):
    """Same call as the reference's `select_model(model_name, basedir)`; unknown names raise (the reference `raise`s a
    str, i.e. a TypeError -- a ValueError with the known names is the useful equivalent)."""
    if model_name not in _MODELS:
        raise ValueError(f"Unknown model named {model_name}; known: {sorted(_MODELS)}")
    return _MODELS[model_name](basedir, **kwargs)


# ------------------------------------------------------------------------------------------ checkpoints
def lightning_checkpoint(lit: nn.Module, global_step: int = 0, epoch: int = 0) -> Dict:
    """The part of a Lightning `.ckpt` the reference reads back (`pl_load(path)['state_dict']`, S3/run.py:206-212, and
    `trainer.fit(ckpt_path=...)`): the module's `state_dict` under the reference's key names, on the host."""
    return {"state_dict": {k: v.detach().cpu().clone() for k, v in lit.state_dict().items()},
            "global_step": int(global_step), "epoch": int(epoch), "pytorch-lightning_version": "hosnerf_amd"}


def save_checkpoint(lit: nn.Module, path: str, global_step: int = 0, epoch: int = 0) -> None:
    torch.save(lightning_checkpoint(lit, global_step, epoch), path)


def load_checkpoint(lit: nn.Module, path: str, strict: bool = False):
    """`model.load_state_dict(pl_load(path)['state_dict'], strict=False)` (S3/run.py:206-212): a stage-2 checkpoint fills
    `human.*`, a stage-1 checkpoint `model.*`; called once per file for the stage-3 warm start.  Values are copied INTO
    the flat parameter store (the parameters are views of it), so optimiser state and captured graphs stay valid.
    Returns torch's (missing_keys, unexpected_keys)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    return lit.load_state_dict(sd, strict=strict)
