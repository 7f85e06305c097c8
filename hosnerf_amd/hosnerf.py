"""Stage-3 HOSNeRF renderer: background mip-NeRF-360 (+) human-object branch, composited per ray.

Mirrors the renderer part of `LitMipNeRF360` of the reference's stage 3
(3rd_Complete_HOSNeRF/src/model/mipnerf360/model.py:1501-1629 training_step, and the same block repeated
in progress/test_metrics/allimgs_metrics/free_view at :746-814, :951-1018, :1155-1222, :1358-1425):
the two sub-modules keep the reference's attribute names `model` and `human`, so a stage-3 checkpoint's
`state_dict` keys (`model.mlps.*`, `human.*`) load unchanged (run.py:206-212 warm-start path).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import ops
from .human_nerf import Network, default_cfg
from .mipnerf360 import MipNeRF360


# The two branches of a stage-3 step are independent until the z-merge (and again in the backward pass until the optimiser): the
# background branch is a few dozen large MFMA-bound GEMM launches, the human branch ~250 launches of which many are latency- or
# occupancy-bound (26-joint prologue, the volume decoder's 128-workgroup layers, slab reductions, resampling-sized kernels).
# Issued on ONE stream the small ones leave most of the 256 CUs idle; on two streams they run under the other branch's GEMMs.
# What it took: (1) the join at the end of the backward pass is queued from inside it and names the caller's
# stream explicitly (final callbacks run on the engine's worker thread); (2) human outputs consumed on the main stream are
# record_stream'ed; (3) scratch buffers are keyed by stream (ops._stream_key); (4) the library is built without packed-FP32
# VALU instructions: with them the IPE encoder computed wrong values in lanes 48-63 whenever MFMA waves of the other branch were
# co-resident on its SIMD (Makefile, scripts/stress_victims.py).  HOS_TWO_STREAMS=0 restores the one-stream order.
TWO_STREAMS = os.environ.get("HOS_TWO_STREAMS", "1") != "0"


class _JoinAfterBackward(torch.autograd.Function):
    """Identity on a human-branch output.  The human branch's forward ran on the side stream, so autograd runs its backward
    there too -- including the HIP weight-gradient kernels that write the flat gradient buffer directly, which autograd's own
    end-of-backward synchronisation (leaf AccumulateGrad streams only) does not know about.  This node, executed first in the
    backward pass, queues a callback for the END of that pass: the stream that called `backward()` waits for the side stream."""

    @staticmethod
    def forward(ctx, x, owner):
        ctx.owner = owner
        ctx.main = torch.cuda.current_stream(x.device)        # the stream render() was called on: backward() must be called on it too
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        # NOT torch.cuda.current_stream() inside the callback: final callbacks run on whichever thread retires the last node
        # (the engine's device worker), whose thread-local current stream is not the caller's
        owner, main = ctx.owner, ctx.main
        torch.autograd.Variable._execution_engine.queue_callback(lambda: owner.join_side_stream(main))
        return g, None


class HOSNeRF(nn.Module):
    two_streams = TWO_STREAMS

    def __init__(self, cfg=None, basedir: Optional[str] = None, near_bkg: float = 0.1, far_bkg: float = 1e6):
        super().__init__()
        self._side = {}
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
        dev = batch["rays_o_bkg"].device
        if self.two_streams and dev.type == "cuda":
            cur, side = torch.cuda.current_stream(dev), self.side_stream(dev)
            side.wait_stream(cur)                                  # fork: everything queued so far (inputs, last step's Adam) is visible
            with torch.cuda.stream(side):
                out = self.human(t_rand=t_rand, prologue=prologue, with_cycle=with_cycle, static_cycle=static_cycle, **batch)
            if os.environ.get("HOS_TS_SERIAL_FWD") == "1":         # diagnostic: two streams, but the forward halves do not overlap
                cur.wait_stream(side)
            _, hist = self.model(batch_bkg, 1.0, randomized, is_train, self.near_bkg, self.far_bkg, jitters=jitters)
            cur.wait_stream(side)                                  # join before the z-merge
            # The human outputs were allocated on the side stream and are consumed on this one (z-merge, losses, and -- as saved
            # tensors -- their backward kernels).  Autograd releases a node's saved tensors as soon as the node's backward has
            # been QUEUED; without this the allocator would hand their memory to the next side-stream allocation (the human
            # backward, running concurrently) while the z-merge's backward kernel is still reading it.
            for v in out.values():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    v.record_stream(cur)
            if torch.is_grad_enabled() and out["human_rgbsigma"].requires_grad:
                out["human_rgbsigma"] = _JoinAfterBackward.apply(out["human_rgbsigma"], self)
        else:
            _, hist = self.model(batch_bkg, 1.0, randomized, is_train, self.near_bkg, self.far_bkg, jitters=jitters)
            out = self.human(t_rand=t_rand, prologue=prologue, with_cycle=with_cycle, static_cycle=static_cycle, **batch)
        last = hist[-1]
        rgb, hw, idx_fg, order, zh = ops.merge_composite(
            last["tdist"], last["rgb"], last["density"], out["human_rgbsigma"], out["newsmpl_pts"], out["pts_mask"],
            batch["rays_o_bkg"], batch["rays_d_bkg"], batch["newsmpl_to_scale_world"])
        out.update(rgb=rgb, idx_fg=idx_fg, total_order=order, human_weights_sorted=hw, z_vals_human=zh, ray_history=hist)
        return out

    def side_stream(self, device) -> "torch.cuda.Stream":
        key = torch.device(device).index or 0
        if key not in self._side:
            self._side[key] = torch.cuda.Stream(device)
        return self._side[key]

    def join_side_stream(self, main: Optional["torch.cuda.Stream"] = None):
        """`main` (default: the current stream) waits for everything queued on the human branch's side stream."""
        for s in self._side.values():
            (main if main is not None else torch.cuda.current_stream(s.device)).wait_stream(s)

    def render_bkg_only(self, batch_bkg: Dict[str, torch.Tensor], randomized: bool = False, is_train: bool = False,
                        jitters=None) -> torch.Tensor:
        """Rays that miss the human bounding box (M:818-836, :1434-1452): the last level's 32 samples through the
        NeRF-style composite `_raw2outputs` with an all-ones mask.  Returns rgb [B,3]."""
        _, hist = self.model(batch_bkg, 1.0, randomized, is_train, self.near_bkg, self.far_bkg, jitters=jitters)
        last = hist[-1]
        z = last["tdist"][..., :-1].contiguous()
        rgbsigma = torch.cat([last["rgb"], last["density"][..., None]], -1).contiguous()
        return ops.raw2outputs(rgbsigma, z, batch_bkg["rays_d"].contiguous(), None, None)[0]
