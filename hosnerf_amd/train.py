"""Training-step pieces for the stage-1 (background) model: losses on HIP kernels, the flat fused
Adam, the reference's LR schedule and the ray-sharded data-parallel step.

Mirrors 1st_State-Conditional_Scene/src/model/mipnerf360/model.py:491-514 (training_step),
:536-569 (configure_optimizers / optimizer_step), :611-627 (interlevel / distortion losses) and
run.py:155 (`gradient_clip_val=grad_max_norm`, norm clipping).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import ops
from .flat import FlatModule


def stage1_loss(rgb: torch.Tensor, target: torch.Tensor, ray_history: List[Dict[str, torch.Tensor]],
                data_loss_mult: float = 1.0, interlevel_loss_mult: float = 1.0, distortion_loss_mult: float = 0.01,
                charb_padding: float = 0.001) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """M1:491-514: sqrt(mse + charb^2) + interlevel + 0.01 * distortion (means over the local ray batch).  On the device the tail behind
    the per-ray interlevel / distortion kernels is ONE launch each way (ops.stage1_loss_tail); CPU tensors take the torch form."""
    last = ray_history[-1]
    c, w = last["sdist"], last["weights"]
    B, Sc = w.shape
    if rgb.is_cuda and len(ray_history) <= 3:
        inter = [ops.interlevel_loss_per_ray(c, w, h["sdist"], h["weights"]) for h in ray_history[:-1]]
        dist_ray = ops.distortion_loss_per_ray(c, w)
        total, parts = ops.stage1_loss_tail(rgb, target.to(rgb.dtype), inter, dist_ray, Sc, data_loss_mult, interlevel_loss_mult,
                                            distortion_loss_mult, charb_padding)
        return total, {"mse": parts[1], "interlevel": parts[2], "distortion": parts[3]}
    mse = torch.mean((rgb - target.to(rgb.dtype)) ** 2)
    loss = torch.sqrt(mse + charb_padding**2) * data_loss_mult
    inter = rgb.new_zeros(())
    for h in ray_history[:-1]:
        inter = inter + ops.interlevel_loss_per_ray(c, w, h["sdist"], h["weights"]).sum() / (B * Sc)
    distortion = ops.distortion_loss_per_ray(c, w).mean()
    total = loss + inter * interlevel_loss_mult + distortion * distortion_loss_mult
    return total, {"mse": mse.detach(), "interlevel": inter.detach(), "distortion": distortion.detach()}


def stage1_lr(step: int, max_steps: int, lr_init: float = 2.0e-3, lr_final: float = 2.0e-5,
              lr_delay_steps: int = 512, lr_delay_mult: float = 0.01) -> float:
    """M1:551-563 log-linear decay with a sine warm-up."""
    if lr_delay_steps > 0:
        delay = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * float(np.clip(step / lr_delay_steps, 0, 1)))
    else:
        delay = 1.0
    t = float(np.clip(step / max_steps, 0, 1))
    return delay * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


def allreduce_flat_grad_async(module: FlatModule, group=None, ranges=None) -> list:
    """`allreduce_flat_grad` with the collectives only ENQUEUED (they run on the communicator's stream, ordered after the work
    already on the current stream): returns the work handles; `h.wait()` makes the current stream wait for one.  Used to run
    the volume decoder's backward under the gradient exchange of the other parameters (bench.py, N > 1)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return []
    # every rank must take the same skip / update decision: the range-guard word travels with the gradients here too
    # (4 bytes, MAX, idempotent -- a second exchange of the same step reduces an already agreed word)
    guard = _allreduce_range_guard(module.flat_grad.device, group, async_op=True)
    handles = [guard] if guard is not None else []
    if ranges is None and getattr(module.store, "inactive", None):
        ranges = module.store.active_spans()
    if ranges is None:
        return handles + [dist.all_reduce(module.flat_grad, group=group, async_op=True)]
    return handles + [dist.all_reduce(module.flat_grad[off:off + n], group=group, async_op=True) for off, n in ranges if n > 0]


_HOS_COMM = None


def use_hoscomm(hos):
    """Route the gradient exchange (`allreduce_flat_grad`, the range-guard word) through a `comm.HosComm` communicator -- RCCL
    called from libhoscomm.so on the CURRENT stream, i.e. capturable: with it the whole data-parallel step (forward, backward,
    collectives, optimiser) is ONE hipGraph per rank instead of two or three graphs with eager torch.distributed collectives in
    between (bench.py, HOS_HOSCOMM=1).  `None` restores torch.distributed.  Returns the previous communicator."""
    global _HOS_COMM
    prev, _HOS_COMM = _HOS_COMM, hos
    return prev


def allreduce_flat_grad(module: FlatModule, group=None, ranges=None, hos=None) -> int:
    """Sum the flat gradient over the data-parallel group (RCCL on MI355X, gloo in the CPU tests) and return
    the world size; the 1/world averaging is folded into the Adam kernel's grad_scale.
    ONE collective per step over the whole gradient: xGMI is point-to-point, so a single large message keeps
    every link busy (DDP's default 25 MB buckets would split the 38 MB stage-1 gradient in two).
    `ranges` [(offset, numel), ...] restricts the exchange to those spans (the human network's volume decoder is reduced at
    its 3.5 MB output instead of its 253 MB of parameters, Network.decoder_backward).
    `hos` (default: the communicator set by `use_hoscomm`): the same sums through libhoscomm.so on the current stream."""
    hos = _HOS_COMM if hos is None else hos
    if getattr(module, "decoder_shard", None) is not None and ranges is None:
        raise RuntimeError("this module's volume decoder is sharded over the ranks: exchange `module.reduce_ranges()` only "
                           "(train.backward_human / finish_backward_human do), never the whole flat gradient")
    if hos is not None:
        if ranges is None:
            ranges = module.store.active_spans() if getattr(module.store, "inactive", None) else [(0, module.flat_grad.numel())]
        flag, _ = ops.range_guard_words(module.flat_grad.device) if ops.RANGE_GUARD else (None, None)
        if flag is not None and hos.world > 1:
            hos.all_reduce_max_u32(flag)
        for off, n in ranges:
            if n > 0:
                hos.all_reduce(module.flat_grad[off:off + n], average=False)
        return hos.world
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    world = dist.get_world_size(group)
    if world > 1:
        _allreduce_range_guard(module.flat_grad.device, group)
        if ranges is None and getattr(module.store, "inactive", None):
            ranges = module.store.active_spans()      # never-written spans (all zeros) are not worth exchanging
        if ranges is None:
            dist.all_reduce(module.flat_grad, group=group)
        else:
            for off, n in ranges:
                if n > 0:
                    dist.all_reduce(module.flat_grad[off:off + n], group=group)
    return world


def _allreduce_range_guard(device, group=None, async_op: bool = False):
    """Every rank must take the same skip / update decision in the optimiser launch, or the replicas part for good: the fp16
    range-guard word is MAX-reduced over the group wherever the gradients are exchanged (4 bytes, idempotent; a rank whose rays
    tripped the guard makes every rank skip this step).  Eager only -- like the gradient exchange it sits between the captured
    halves of a multi-rank step.  `async_op`: returns the work handle (None when there is nothing to reduce)."""
    if device.type != "cuda" or not ops.RANGE_GUARD or torch.cuda.is_current_stream_capturing():
        return None
    flag, _ = ops.range_guard_words(device)
    if flag is not None:
        return dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group, async_op=async_op)
    return None


class ShardComm:
    """The collectives of a module whose parameters are SHARDED over the data-parallel ranks (the human network's volume decoder,
    `Network.shard_decoder`): sum-all-reduce and all-gather of small fp32 tensors on the current stream.  With `hos` (a
    `comm.HosComm`: RCCL through libhoscomm.so) the calls are plain stream work and may sit inside a captured hipGraph -- the
    decoder's forward is the first thing of the human branch, i.e. in the middle of the captured step; without it they go through
    torch.distributed (`group`; any backend, eager only -- the CPU-side harness of the tests runs two ranks on one GPU over gloo)."""

    def __init__(self, rank: int, world: int, group=None, hos=None):
        self.rank, self.world, self.group, self.hos = int(rank), int(world), group, hos
        if hos is not None and (hos.rank != self.rank or hos.world != self.world):
            raise ValueError("ShardComm: the HosComm communicator has another rank / size")

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return t
        if self.hos is not None:
            return self.hos.all_reduce(t, average=False)
        dist.all_reduce(t, group=self.group)
        return t

    def all_gather(self, send: torch.Tensor) -> torch.Tensor:
        """[world, *send.shape]: rank r's tensor at index r."""
        send = send.contiguous()
        if self.world == 1:
            return send.unsqueeze(0)
        if self.hos is not None:
            return self.hos.all_gather(send)
        recv = torch.empty((self.world,) + tuple(send.shape), device=send.device, dtype=send.dtype)
        dist.all_gather(list(recv.unbind(0)), send, group=self.group)
        return recv


def _shard_norm_correction(opts, target: torch.Tensor):
    """A rank that owns a SHARD of some parameters has only that shard's gradient: its local sum of squares is R + s_r (R: the
    replicated parameters, identical on every rank; s_r: its shard).  The norm the reference clips by is over ALL parameters,
    R + sum_r s_r: add (sum_r s_r - s_r) to the first element of `target` (the partial-sum array / the scalar the Adam launch
    adds up).  One 4-byte all-reduce per step; a no-op without sharded modules."""
    for o in opts:
        comm = getattr(o.module, "decoder_shard", None)
        if comm is None or comm.world == 1:
            continue
        g = o.module.flat_grad
        bufs = getattr(o.module, "_shard_norm_bufs", None)
        if bufs is None or bufs.device != g.device:
            bufs = o.module._shard_norm_bufs = torch.zeros(2, device=g.device)
        ops.zero_(bufs)
        for off, n in o.module.decoder_shard_spans():
            ops.sumsq(g[off:off + n], bufs[0:1])
        ops.copy_or_zero_n([bufs[1:2]], [bufs[0:1]])
        comm.all_reduce_sum(bufs[1:2])
        target[0:1].add_(bufs[1:2] - bufs[0:1])


class GradClip:
    """`Trainer(gradient_clip_val=max_norm, gradient_clip_algorithm="norm")` of the reference's launchers (S1/run.py:155,
    2nd_.../run.py:185-186, 3rd_.../run.py:188-189; every Backpack.gin binds `run.grad_max_norm = 0.001`): Lightning calls
    `clip_grad_norm_` on ALL parameters of the step's optimiser, i.e. ONE norm -- in stage 3 over the background model AND the
    human network, which live in two flat buffers here.  The object owns the device scalar the sum of squares of every
    participating flat gradient is accumulated into (one `hos_sumsq` launch per buffer); the Adam launches of all
    participants read it and scale their gradients by min(max_norm / (norm + 1e-6), 1)."""

    def __init__(self, max_norm: float):
        self.max_norm = float(max_norm)
        self._buf = None

    def sumsq(self, opts) -> Optional[torch.Tensor]:
        """Sum of squares of the (already reduced) flat gradients of `opts` in one device scalar; None when clipping is off."""
        if self.max_norm <= 0:
            return None
        dev = opts[0].module.flat_grad.device
        if self._buf is None or self._buf.device != dev:
            self._buf = torch.zeros(1, device=dev)
        self._buf.zero_()
        for o in opts:
            g = o.module.flat_grad
            for off, n in o.module.store.active_spans():          # the whole buffer unless the module has inactive spans
                ops.sumsq(g[off:off + n], self._buf)
        _shard_norm_correction(opts, self._buf)
        return self._buf

    def partials(self, opts) -> Optional[torch.Tensor]:
        """Round 4: the same norm as per-block partial sums of ALL participating spans in ONE launch (hos_sumsq_partials: fixed
        order, no atomics, nothing to zero first); `ops.adam_multi` adds them.  None when clipping is off or a span is unaligned."""
        if self.max_norm <= 0:
            return None
        spans = []
        for o in opts:
            g = o.module.flat_grad
            spans += [g[off:off + n] for off, n in o.module.store.active_spans()]
        if len(spans) > 32 or any((t.numel() % 4) or (t.data_ptr() % 16) for t in spans):
            return None
        dev = spans[0].device
        if getattr(self, "_partials", None) is None or self._partials.device != dev:
            self._partials = torch.empty(ops.sumsq_blocks(), device=dev)
        ops.sumsq_partials(spans, self._partials)
        _shard_norm_correction(opts, self._partials)
        return self._partials


def _minus_inactive(lr_ranges, module: FlatModule):
    """Learning-rate ranges with the store's inactive spans cut out (parameters whose gradient is identically zero by
    construction: Adam leaves them where they are, so they are neither read nor written)."""
    inactive = sorted(getattr(module.store, "inactive", []))
    if not inactive:
        return lr_ranges
    ranges = lr_ranges or [(0, module.flat_param.numel(), 1.0)]
    out = []
    for off, n, mult in ranges:
        pos, end = off, off + n
        for io, inn in inactive:
            if io + inn <= pos or io >= end:
                continue
            if io > pos:
                out.append((pos, io - pos, mult))
            pos = max(pos, io + inn)
        if pos < end:
            out.append((pos, end - pos, mult))
    return out


LAZY_ADAM = __import__("os").environ.get("HOS_LAZY_ADAM", "1") != "0"     # A/B switch: 0 = every span updated every step (rounds 1-5)


def _split_at_lazy(ranges, lazy_spans):
    """Cut the learning-rate ranges [(off, n, mult)] at the lazily updated spans [(off, n)] (sorted; each inside one range, float4-
    aligned).  A RUN of adjacent spans of equal length inside one range (the [n_states, 64] block of state embeddings) stays ONE range
    with a row length: returns (ranges, per range None | (index of its first lazy span, number of rows, row length))."""
    runs, i = [], 0
    while i < len(lazy_spans):
        lo, ln = lazy_spans[i]
        k = 1
        while i + k < len(lazy_spans) and lazy_spans[i + k] == (lo + k * ln, ln):
            k += 1
        runs.append((lo, ln, i, k))
        i += k
    out, idx = [], []
    for off, n, mult in ranges:
        pos, end = off, off + n
        pending = []
        for lo, ln, first, k in runs:
            if lo + k * ln <= pos or lo >= end:
                continue
            if lo < pos or lo % 4 or ln % 4:
                raise ValueError(f"lazy span ({lo}, {ln}) straddles a learning-rate range or is not float4-aligned")
            if lo + k * ln > end:                          # a run may end with the range only on a span boundary
                fit = (end - lo) // ln
                if fit <= 0 or lo + fit * ln != end:
                    raise ValueError(f"lazy span ({lo}, {ln}) straddles a learning-rate range or is not float4-aligned")
                pending.append((lo + fit * ln, ln, first + fit, k - fit))
                k = fit
            if lo > pos:
                out.append((pos, lo - pos, mult)); idx.append(None)
            out.append((lo, k * ln, mult)); idx.append((first, k, ln))
            pos = lo + k * ln
        runs = [r for r in runs if not (off <= r[0] < end)] + pending
        runs.sort()
        if pos < end:
            out.append((pos, end - pos, mult)); idx.append(None)
    return out, idx


class FusedAdam:
    """torch.optim.Adam semantics over the flat parameter buffer of a FlatModule: one sum-of-squares
    launch (norm clipping), one RCCL all-reduce of the whole gradient (multi-GPU) and one Adam launch."""

    def __init__(self, module: FlatModule, lr: float = 2e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 max_grad_norm: float = 0.0, process_group=None, lr_ranges=None, clip: Optional[GradClip] = None):
        """`lr_ranges`: optional [(offset, numel, lr_multiplier), ...] over the flat buffer for per-module learning
        rates (the reference's name-keyed param groups, core/train/optimizers/human_nerf/optimizer.py:19-60).
        `clip`: a GradClip SHARED with the other FusedAdams of the same step (stage 3: one global norm over both modules,
        applied by `step_all`); `max_grad_norm` alone gives this optimiser its own."""
        self.module = module
        self.lr_ranges = _minus_inactive(lr_ranges, module)
        self.lr, self.betas, self.eps = lr, betas, eps
        # Lazily updated spans (round 6; hos_adam_lazy_prepare): torch's Adam -- the reference's, under Lightning's zero_grad(set_to_none) --
        # SKIPS a parameter whose gradient is None and counts that parameter's own steps.  The modules name the spans that can be
        # without a gradient in a step (`lazy_param_spans`: state embeddings of the other states, the pose decoder before its kick-in);
        # each becomes a range of its own with a device state row {t, active, 1-b1^t, 1/sqrt(1-b2^t), scratch}.
        self.lazy_spans = sorted(module.lazy_param_spans()) if (LAZY_ADAM and hasattr(module, "lazy_param_spans")) else []
        self._range_lazy = None
        if self.lazy_spans:
            self.lr_ranges, self._range_lazy = _split_at_lazy(self.lr_ranges or [(0, module.flat_param.numel(), 1.0)], self.lazy_spans)
        self.lazy_state = torch.zeros(len(self.lazy_spans), 8, device=module.flat_param.device)     # rows of hos_adam_lazy_prepare
        self.clip = clip if clip is not None else GradClip(max_grad_norm)
        self.group = process_group
        p = module.flat_param
        self.exp_avg = torch.zeros_like(p)
        self.exp_avg_sq = torch.zeros_like(p)
        self.step_count = 0
        self.grad_is_reduced = False     # set by `backward_human`: the next step must not all-reduce the flat gradient again
        self._hyper = None      # device [lr, 1-b1^t, 1/sqrt(1-b2^t)] for graph-captured steps

    def set_step_hyper(self, lr: Optional[float] = None):
        """Advance the step counter and refresh the device hyper-parameter block (call OUTSIDE a captured graph, before
        each replay): one row [lr * multiplier, 1-b1^t, 1/sqrt(1-b2^t)] per learning-rate range.  `step(dynamic=True)`
        then reads lr / bias corrections from device memory."""
        self.step_count += 1
        lr = self.lr if lr is None else lr
        ranges = self.lr_ranges or [(0, 0, 1.0)]
        bc1, bc2 = 1.0 - self.betas[0] ** self.step_count, 1.0 / math.sqrt(1.0 - self.betas[1] ** self.step_count)
        if self._hyper is None:
            self._hyper = torch.empty(len(ranges), 3, device=self.module.flat_param.device)
            # a RING of pinned staging buffers: the copy below is asynchronous and the host runs many replays ahead of the GPU,
            # so rewriting ONE staging buffer would hand an earlier step the hyper-parameters of a later one.  A slot is reused
            # only after the copy that read it has executed (its event), which also bounds the run-ahead to the ring size.
            self._hyper_ring = [torch.empty(len(ranges), 3, dtype=torch.float32).pin_memory() for _ in range(8)]
            self._hyper_events = [None] * len(self._hyper_ring)
        slot = self.step_count % len(self._hyper_ring)
        if self._hyper_events[slot] is not None:
            self._hyper_events[slot].synchronize()
        host = self._hyper_ring[slot]
        for r, (_, _, mult) in enumerate(ranges):
            host[r, 0], host[r, 1], host[r, 2] = lr * mult, bc1, bc2
        self._hyper.copy_(host, non_blocking=True)
        if self._hyper.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self._hyper.device))
            self._hyper_events[slot] = ev

    def zero_grad(self):
        self.module.store.zero_grad()

    def world_size(self) -> int:
        if _HOS_COMM is not None:
            return _HOS_COMM.world
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    @property
    def max_grad_norm(self) -> float:
        return self.clip.max_norm

    def step(self, lr: Optional[float] = None, dynamic: bool = False, reduced: bool = False, clip_sumsq=False, clear_guard: bool = True):
        """`reduced=True`: the caller already all-reduced the flat gradient (e.g. overlapped with other work); only the
        1/world scaling is applied here.  `clip_sumsq`: the joint sum of squares `step_all` formed over every optimiser of
        the step (a device scalar, or None = no clipping); left at False this optimiser clips by its own norm.
        `clear_guard=False`: this is NOT the last optimiser launch of the training step -- the fp16 range-guard word is consumed
        (a poisoned step is skipped) but left set for the launches that follow (`step_all` passes it).  A loop that
        steps several optimisers itself must do the same, or call `step_all`."""
        self.module.store.ensure_bound()          # torch autograd's prologue gradients must have landed in the flat buffer
        _no_pending_decoder_backward(self.module)
        reduced, self.grad_is_reduced = reduced or self.grad_is_reduced, False
        g = self.module.flat_grad
        if self.exp_avg.device != g.device:       # the module was moved after the optimiser was built (`lit.to(device)`)
            self.exp_avg, self.exp_avg_sq, self.lazy_state = self.exp_avg.to(g.device), self.exp_avg_sq.to(g.device), self.lazy_state.to(g.device)
            self._hyper = None
        world = self.world_size() if reduced else allreduce_flat_grad(self.module, self.group)
        if clip_sumsq is False and _step_multi([self], [lr], dynamic, clear_guard):         # norm + Adam of this module as two launches
            return
        if clip_sumsq is False and _guard_on_fallback_path(g.device, dynamic):
            return
        sumsq = self.clip.sumsq([self]) if clip_sumsq is False else clip_sumsq      # after the exchange: the norm of the SUMMED gradient
        if self.lazy_spans and not getattr(self, "_lazy_warned", False):
            import warnings          # (HOS_MULTI_ADAM=0, > 32 spans, unaligned ranges: the per-span launches know no lazily updated spans)
            warnings.warn("FusedAdam: per-span Adam path taken -- parameters without a gradient are updated with zero gradients here "
                          "(torch's Adam would skip them; the one-launch path does)")
            self._lazy_warned = True
        if dynamic:
            p = self.module.flat_param
            for r, (off, n, _) in enumerate(self.lr_ranges or [(0, p.numel(), 1.0)]):
                ops.adam_step_dyn(p[off:off + n], g[off:off + n], self.exp_avg[off:off + n], self.exp_avg_sq[off:off + n],
                                  self._hyper[r], self.betas[0], self.betas[1], self.eps, 1.0 / world, sumsq, self.max_grad_norm)
            return
        self.step_count += 1
        lr = self.lr if lr is None else lr
        p = self.module.flat_param
        for off, n, mult in (self.lr_ranges or [(0, p.numel(), 1.0)]):
            ops.adam_step(p[off:off + n], g[off:off + n], self.exp_avg[off:off + n], self.exp_avg_sq[off:off + n], lr * mult,
                          self.betas[0], self.betas[1], self.eps, self.step_count, 1.0 / world, sumsq, self.max_grad_norm)

    def state_dict(self):
        """Flat layout (version 1): the two moment buffers in the order of the module's flat parameter buffer."""
        return {"layout": "hosnerf_amd.flat.v1", "numel": int(self.exp_avg.numel()), "exp_avg": self.exp_avg.detach().cpu().clone(),
                "exp_avg_sq": self.exp_avg_sq.detach().cpu().clone(), "step": self.step_count, "lr": self.lr,
                "lazy_spans": list(self.lazy_spans), "lazy_state": self.lazy_state.detach().cpu().clone()}

    def load_state_dict(self, sd):
        if sd.get("layout", "hosnerf_amd.flat.v1") != "hosnerf_amd.flat.v1" or int(sd.get("numel", self.exp_avg.numel())) != self.exp_avg.numel():
            raise ValueError("optimizer state was saved for a different flat parameter layout")
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count = int(sd["step"])
        if len(self.lazy_spans):
            if [tuple(x) for x in sd.get("lazy_spans", [])] == [tuple(x) for x in self.lazy_spans] and tuple(sd["lazy_state"].shape) == tuple(self.lazy_state.shape):
                self.lazy_state.copy_(sd["lazy_state"])
                self.lazy_state[:, 4:] = 0
            else:       # a checkpoint written before round 6: every span was updated at every step
                t = float(self.step_count)
                self.lazy_state[:, 0] = t
                self.lazy_state[:, 1] = 1.0
                self.lazy_state[:, 2] = 1.0 - self.betas[0] ** max(t, 1.0)
                self.lazy_state[:, 3] = 1.0 / math.sqrt(1.0 - self.betas[1] ** max(t, 1.0))


MULTI_ADAM = __import__("os").environ.get("HOS_MULTI_ADAM", "1") != "0"     # A/B switch: 0 = one norm / Adam launch per span (round 3)


def _step_multi(opts, lrs, dynamic: bool, clear_guard: bool = True) -> bool:
    """The whole optimiser step of `opts` (already reduced gradients) as TWO launches: the joint gradient norm of every active span
    (hos_sumsq_partials) and torch.optim.Adam over every learning-rate range of every module (hos_adam_multi), which also
    consumes the fp16 range-guard word: a step whose forward left the exact hi/lo range updates nothing (train.range_skips counts).
    Returns False when the step does not fit that form (CPU tensors, > 16 spans, unaligned ranges, separate clip objects)."""
    if not MULTI_ADAM or not opts or not opts[0].module.flat_param.is_cuda:
        return False
    clip = opts[0].clip
    if not all(o.clip is clip for o in opts):
        return False
    world = opts[0].world_size()
    spans = []
    for o, l in zip(opts, lrs):
        p, g = o.module.flat_param, o.module.flat_grad
        if o.exp_avg.device != g.device:
            o.exp_avg, o.exp_avg_sq, o.lazy_state = o.exp_avg.to(g.device), o.exp_avg_sq.to(g.device), o.lazy_state.to(g.device)
            o._hyper = None
        if dynamic and o._hyper is None:
            return False
        l = o.lr if l is None else l
        for r, (off, n, mult) in enumerate(o.lr_ranges or [(0, p.numel(), 1.0)]):
            if n % 4 or off % 4:
                return False
            k = o._range_lazy[r] if o._range_lazy is not None else None
            spans.append((p[off:off + n], g[off:off + n], o.exp_avg[off:off + n], o.exp_avg_sq[off:off + n],
                          o._hyper[r] if dynamic else None, float(l) * mult, None if k is None else o.lazy_state[k[0]:k[0] + k[1]],
                          0 if (k is None or k[1] == 1) else k[2]))
    if len(spans) > 32 or any(o.betas != opts[0].betas or o.eps != opts[0].eps for o in opts):
        return False
    partial = clip.partials(opts)
    if clip.max_norm > 0 and partial is None:
        return False
    if not dynamic:
        for o in opts:
            o.step_count += 1
        if any(o.step_count != opts[0].step_count for o in opts):
            for o in opts:
                o.step_count -= 1
            return False
    dev = spans[0][0].device
    guard = ops.range_guard_words(dev)       # with several ranks the word was MAX-reduced next to the gradients (allreduce_flat_grad)
    lg, ls = [], []
    for sp in spans:                         # one (gradient slice, state row) per lazily updated span -- a row-span contributes its rows
        if sp[6] is not None:
            rows = sp[6].shape[0]
            rl = sp[1].numel() // rows
            lg += [sp[1][i * rl:(i + 1) * rl] for i in range(rows)]
            ls += [sp[6][i] for i in range(rows)]
    if lg:                                   # which of them took part in this step (their reduced gradient is not all zero)
        ops.adam_lazy_prepare(lg, ls, opts[0].betas[0], opts[0].betas[1], guard[0])
    if not clear_guard:                      # an earlier launch of a multi-launch step: skip on the word, leave it (and the count) alone
        guard = (guard[0], None)
    ops.adam_multi(spans, opts[0].step_count if not dynamic else 0, opts[0].betas[0], opts[0].betas[1], opts[0].eps, 1.0 / world,
                   partial, clip.max_norm, guard)
    return True


_GUARDLESS_WARNED = False


def _guard_on_fallback_path(device, dynamic: bool) -> bool:
    """The per-span Adam launches (hos_adam_step / hos_adam_step_dyn: > 16 spans, unaligned ranges, separate clip objects,
    HOS_MULTI_ADAM=0) take no guard word.  Outside a graph capture the host reads the flag itself (one 4-byte read per step of this
    rarely taken path) and returns True when the step must be skipped; under capture that is impossible and the guard is off for
    this optimiser -- said once, not silently."""
    global _GUARDLESS_WARNED
    if not ops.RANGE_GUARD or not torch.cuda.is_available() or device.type != "cuda":
        return False
    if torch.cuda.is_current_stream_capturing():
        if not _GUARDLESS_WARNED:
            import warnings
            warnings.warn("optimizer step captured on the per-span Adam path: the fp16 range guard does not cover it "
                          "(poll train.check_range from the loop)")
            _GUARDLESS_WARNED = True
        return False
    return bool(ops.range_events(device, reset=True))


def range_skips(device, since_last_poll: bool = False) -> int:
    """Number of optimiser steps the device-side range guard has skipped (one 4-byte read: poll it rarely)."""
    return ops.range_skips(device, since_last_poll)


def _no_pending_decoder_backward(module):
    """A human network whose forward cut the graph at the motion-weight volume (`split_decoder_backward`) has the decoder's
    backward still pending after `loss.backward()`: stepping now would train every module but the decoder, silently.
    `train.backward_human` / `finish_backward_human` / `Network.finish_decoder_backward` run it."""
    pend = getattr(module, "pending_volume_grad", None)
    if pend is not None and pend() is not None:
        raise RuntimeError("optimizer step with the volume decoder's backward still pending: call train.finish_backward_human(net, opt) "
                           "(or net.finish_decoder_backward()) after loss.backward(), or leave net.split_decoder_backward False")


def step_all(opts, lr=None, dynamic: bool = False, reduced=False):
    """The optimiser step of a training step that owns several flat modules (stage 3: background + human; the reference has
    ONE torch Adam over both, optimizer.py:19-60, and Lightning clips ONE norm over it): (1) every flat gradient that was
    not exchanged yet is all-reduced, (2) the joint sum of squares of the reduced gradients is formed once (`GradClip`,
    shared by the optimisers), (3) every Adam launch scales by the same clip coefficient.  `lr`: a float or one per optimiser;
    `reduced`: a bool or one per optimiser."""
    opts = list(opts)
    lrs = list(lr) if isinstance(lr, (list, tuple)) else [lr] * len(opts)
    red = list(reduced) if isinstance(reduced, (list, tuple)) else [reduced] * len(opts)
    for o, r in zip(opts, red):
        o.module.store.ensure_bound()
        _no_pending_decoder_backward(o.module)
        if not (r or o.grad_is_reduced):
            allreduce_flat_grad(o.module, o.group)
        o.grad_is_reduced = False
    if _step_multi(opts, lrs, dynamic):
        return
    if _guard_on_fallback_path(opts[0].module.flat_param.device, dynamic):
        return
    clip = opts[0].clip
    shared = all(o.clip is clip for o in opts)
    for o, l in zip(opts, lrs):
        if shared:
            if o is opts[0]:
                ss = clip.sumsq(opts)
            o.step(l, dynamic=dynamic, reduced=True, clip_sumsq=ss)
        else:
            o.step(l, dynamic=dynamic, reduced=True, clear_guard=o is opts[-1])     # ONE skip decision per training step


class FusedAdamOptimizer(torch.optim.Optimizer):
    """`torch.optim.Optimizer` face of one or more FusedAdams -- what `configure_optimizers` hands to a Lightning-style loop
    (the reference builds ONE torch Adam over both stage-3 modules with name-keyed learning rates, optimizer.py:19-60).
    One param group per flat module; `zero_grad()` (whatever `set_to_none` says) zeroes the flat gradient buffers and keeps
    every `p.grad` aliased to them, `step()` runs the one-launch Adam of each module with its group's current `lr`
    (schedulers / `optimizer_step` hooks write `param_groups[i]['lr']` like they do for torch's Adam), `state_dict()`
    carries the moments in the flat layout."""

    def __init__(self, modules, **kw):
        mods = list(modules) if isinstance(modules, (list, tuple)) else [modules]
        self.fused = [m if isinstance(m, FusedAdam) else FusedAdam(m, **kw) for m in mods]
        groups = [{"params": list(f.module.parameters()), "lr": f.lr, "name": type(f.module).__name__} for f in self.fused]
        super().__init__(groups, dict(lr=self.fused[0].lr, betas=self.fused[0].betas, eps=self.fused[0].eps))

    def zero_grad(self, set_to_none: bool = False):
        for f in self.fused:
            f.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        step_all(self.fused, [float(g["lr"]) for g in self.param_groups])
        return loss

    def state_dict(self):
        return {"fused": [f.state_dict() for f in self.fused], "lr": [float(g["lr"]) for g in self.param_groups]}

    def load_state_dict(self, sd):
        for f, fsd in zip(self.fused, sd["fused"]):
            f.load_state_dict(fsd)
        for g, lr in zip(self.param_groups, sd.get("lr", [])):
            g["lr"] = float(lr)


def shard_rays(batch: Dict[str, torch.Tensor], rank: int, world: int) -> Dict[str, torch.Tensor]:
    """S1/src/data/sampler.py:96: each rank renders rays[rank::world] of the global batch."""
    if world == 1:
        return batch
    out = {}
    for k, v in batch.items():
        out[k] = v[rank::world].contiguous() if isinstance(v, torch.Tensor) and v.dim() > 0 else v
    return out


HOST_KEYS = ("time", "times", "iter_val", "is_train", "img_width", "img_height", "frame_name", "mse_const", "mse_count")


def batch_to_device(batch: Dict, device) -> Dict:
    """Move a dataset item to the device EXCEPT the control scalars the networks read on the host (`time` -> state
    selection and the flow switch, `iter_val` -> kick-in switches and the hann window, N:589-656).  The reference moves them
    too (`cpu_data_to_gpu`, M:1507) and then reads each back with an implicit `.item()`: a device->host round trip at the
    START of every step, which keeps the host from queueing step n+1 while step n still runs."""
    return {k: (v.to(device, non_blocking=True) if isinstance(v, torch.Tensor) and k not in HOST_KEYS else v) for k, v in batch.items()}


def shard_frame(n_rays: int, rank: int, world: int):
    """Inference partition of a frame's rays: contiguous ranges, every rank the same length -- like the
    reference (S1/src/data/interface.py:152-166) the tail is padded by repeating the last rays so that the all-gather
    is regular.  Returns (index tensor of this rank's rays [per], per)."""
    per = (n_rays + world - 1) // world
    idx = torch.arange(rank * per, (rank + 1) * per).clamp_(max=n_rays - 1)
    return idx, per


def gather_frame(rgb_local: torch.Tensor, n_rays: int, group=None) -> torch.Tensor:
    """All-gather of the per-rank RGB [per, 3] into the frame's [n_rays, 3] (the reference's `alter_gather_cat`,
    S1/src/model/interface.py:30-39, minus its padding rows).  ONE collective per frame (24.9 MB at 1080p)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return rgb_local[:n_rays]
    world = dist.get_world_size(group)
    out = torch.empty(world * rgb_local.shape[0], *rgb_local.shape[1:], dtype=rgb_local.dtype, device=rgb_local.device)
    dist.all_gather_into_tensor(out, rgb_local.contiguous(), group=group)
    return out[:n_rays]


def train_step_stage1(model, opt: FusedAdam, batch: Dict[str, torch.Tensor], train_frac: float, near: float,
                      far: float, lr: Optional[float] = None):
    """One full stage-1 optimisation step: forward (3 levels) + losses + backward + clip + Adam."""
    opt.zero_grad()
    renderings, hist = model(batch, train_frac, True, True, near, far)
    loss, parts = stage1_loss(renderings[-1]["rgb"], batch["target"], hist)
    loss.backward()
    opt.step(lr)
    return loss.detach(), parts


# ------------------------------------------------------------------------------------------ stage 3 (C4)
def human_lr_ranges(net, lr_cnl: float = 6.667e-5, lr_other: float = 6.667e-6):
    """configs/default.yaml train.lr_*: cnl_mlp and human_stateembeds train at lr_cnl, every other human module
    (mweight_vol_decoder, pose_decoder, non_rigid_mlp, non_rigid_forward_mlp) at lr_other.  The flat layout puts
    the canonical MLP and the state embeddings last, so two contiguous ranges cover it (multipliers of lr_cnl)."""
    start = net._cnl[0].W.offset
    start -= start % 4
    return [(0, start, lr_other / lr_cnl), (start, net.flat_param.numel() - start, 1.0)]


def lr_ranges_by_name(module: FlatModule, lr_of, base_lr: float):
    """Per-parameter learning rates as contiguous ranges of the flat buffer: `lr_of(name)` is the learning rate of the
    parameter `name` (the reference builds one Adam param group per parameter, keyed by the first `cfg.train.lr_<key>`
    whose key occurs in the name, optimizer.py:19-60); returns [(offset, numel, lr / base_lr), ...] with neighbouring
    regions of equal rate merged (regions start on 64-float boundaries, so every range is float4-aligned)."""
    names = {id(p): n for n, p in module.named_parameters()}
    spans = sorted({r.offset: names[id(p)] for p, r, _, _ in module.store._bindings}.items())
    total = module.flat_param.numel()
    out = []
    for i, (off, name) in enumerate(spans):
        end = spans[i + 1][0] if i + 1 < len(spans) else total
        mult = float(lr_of(name)) / base_lr
        if out and abs(out[-1][2] - mult) < 1e-12 * max(1.0, abs(mult)):
            out[-1] = (out[-1][0], end - out[-1][0], out[-1][2])
        else:
            out.append((off, end - off, mult))
    if out and out[0][0] != 0:
        out[0] = (0, out[0][0] + out[0][1], out[0][2])
    return out


def human_lr_from_cfg(cfg, default_lr: float):
    """(base learning rate, name -> learning rate) of the human network from `cfg.train` (configs/default.yaml: `lr` and the
    `lr_<module>` keys; stage 3 also `lr_bkgd`).  Missing keys fall back to the shipped defaults: the canonical MLP and the
    state embeddings at the base rate, every other module at a tenth of it."""
    tr = getattr(cfg, "train", None) or {}
    get = (lambda k, d: tr.get(k, d)) if isinstance(tr, dict) else (lambda k, d: getattr(tr, k, d))
    base = float(get("lr_cnl_mlp", get("lr", default_lr)))
    keys = ("cnl_mlp", "human_stateembeds", "mweight_vol_decoder", "pose_decoder", "non_rigid_mlp", "non_rigid_forward_mlp")
    rates = {k: float(get("lr_" + k, base if k in ("cnl_mlp", "human_stateembeds") else base / 10.0)) for k in keys}

    def lr_of(name: str) -> float:
        for k in keys:
            if k in name:
                return rates[k]
        return float(get("lr", base))
    return base, lr_of


def prepare_patch_targets(batch: Dict) -> Dict:
    """Host-side data preparation (what the dataset knows when it builds the item, S2/core/data/human_nerf/train.py:585-586):
    `target_rgbs` [N_rays,3] = the target colours of the sampled rays, and the two constants of the patch MSE --
    `_unpack_imgs` (M2:41-50) fills the patch pixels outside the ray mask with the background colour, so they contribute
    `mse_const` = sum |bgcolor/255 - target|^2 over those pixels to the numerator and every patch pixel to `mse_count`.
    Runs on CPU tensors before `batch_to_device`, so the step itself needs no boolean indexing (no host round trip)."""
    if "patch_masks" not in batch:
        return batch
    pm = batch["patch_masks"].bool()
    tp = batch["target_patches"]
    out = dict(batch)
    if "target_rgbs" not in out:
        out["target_rgbs"] = tp[pm]
    bg = (batch["bgcolor"].to(tp.dtype) / 255.0).expand(tp.shape)
    out["mse_const"] = float(((bg - tp) ** 2)[~pm].sum())
    out["mse_count"] = float(tp.numel())
    return out


def _loss_parts(parts: torch.Tensor) -> Dict[str, torch.Tensor]:
    return {"mse": parts[1], "flow": parts[2], "cycle": parts[3]}


def _lpips_term(lpips, rgb, batch, w_lpips: float):
    """`w_lpips * mean_i LPIPS(2 unpack(rgb)_i - 1, 2 target_i - 1)` (M:1673-1676; hosnerf_amd/lpips.py).  The index of every patch
    pixel's ray is cached in the batch (`patch_ray_idx`: built once per item from `patch_masks`, outside a captured step)."""
    from .lpips import patch_ray_index
    if "patch_ray_idx" not in batch:
        batch["patch_ray_idx"] = patch_ray_index(batch["patch_masks"].to(rgb.device))
    return w_lpips * lpips.loss(rgb, batch["target_patches"], batch["patch_ray_idx"], batch["bgcolor"])


def _lpips_named(term, w_lpips: float):
    """The unweighted LPIPS value for the loss report (w_lpips = 0 switches the term off: report 0, do not divide)."""
    return term.detach() / w_lpips if w_lpips != 0 else term.detach() * 0.0


def stage3_losses(out: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor], w_mse: float = 0.2,
                  w_flow: float = 0.01, w_cycle: float = 0.01, lpips=None, w_lpips: float = 1.0):
    """M:1690-1716 `get_loss`: 0.2*MSE + 0.01*flow + 0.01*cycle (configs/default.yaml lossweights) -- one HIP launch
    (hos_train_losses_fwd), gradients in another -- plus, when a loaded `lpips.LPIPS` module is passed, 1.0 * LPIPS on the
    unpacked patches (its ImageNet VGG-16 filters do not exist offline, so the default is without it, on both sides of every
    comparison).  Returns (total, {name: unweighted term}).

    The reference selects the foreground rows first (`ray_grid[idx_fg]`, `human_weights_onlyfg`, M:1704) -- a boolean
    index, i.e. a device->host round trip per step, next to its four `.item()` reads (M:1617-1622).  The kernel runs over
    ALL rays with the foreground flag folded into the flow mask: rows of background rays have zero composite weight in
    `human_weights_sorted`, so the numerator is unchanged, and the denominator sum(M) over the selected [B_fg, S, 1] block
    is S * sum_fg(M_ray).  No synchronisation, fixed shapes; the cycle set's row count may live on the device."""
    rgb = out["rgb"]
    # The reference compares the plain reshape of the rendered rays with `target_patches` (M:1598-1602, `_unpack_imgs` M:41-50) -- NOT
    # with `target_rgbs`.  The two differ where a patch leaves the subject's box: stage 3 keeps the patch whole ("to keep the patch size",
    # core/data/human_nerf/train.py:322-330), a patch pixel outside the box is rendered by the PREVIOUS box ray (cumsum(ray_mask) - 1) and
    # compared with the PATCH pixel's colour, while `target_rgbs` holds the colour of that substituted ray's own pixel.  (Round 6: this
    # used `target_rgbs` when the item carried it -- found by tests/test_gpu_convergence.py on a frame whose patch crossed the box.)
    # Items whose patches are CUT (the synthetic items of tests / bench sweeps, e.g. 512 rays of one 32 x 32 patch: not a stage-3 item of the
    # reference, whose launcher path refuses them) keep the masked unpack of stage 2: their `target_rgbs` are the unmasked patch pixels.
    tp = batch.get("target_patches")
    target = tp.reshape(-1, 3) if (tp is not None and tp.numel() == rgb.numel()) else batch["target_rgbs"]
    flow = "deform_pts_prev_final" in out and "ray_grid" in batch                 # time > 0.005 and training
    total, parts = ops.train_losses(
        rgb, target, batch.get("mse_const", 0.0), batch.get("mse_count"),
        pts_prev=out["deform_pts_prev_final"] if flow else None, weights=out["human_weights_sorted"] if flow else None,
        ray_grid=batch.get("ray_grid"), fg=out["idx_fg"], cam_prev=batch.get("newsmpl_to_camera_prev"),
        intrinsics_prev=batch.get("intrinsics_prev"), observe=out["observe_pts"], deform=out["deform_pts_final"],
        n_cyc_dev=out.get("cycle_count"), w_mse=w_mse, w_flow=w_flow, w_cycle=w_cycle)
    named = _loss_parts(parts)
    if lpips is not None:
        if "patch_ray_idx" not in batch and not bool(batch["patch_masks"].all()):
            # the stage-3 reference unpacks with a plain reshape (M:1673: rgbs.reshape([b, w, h, 3])), i.e. it assumes whole
            # patches; the masked gather below equals it only then (checked once per item, outside a captured step)
            raise ValueError("stage-3 LPIPS: patch_masks with holes -- the reference's stage-3 _unpack_imgs is a plain reshape")
        term = _lpips_term(lpips, rgb, batch, w_lpips)
        total = total + term
        named["lpips"] = _lpips_named(term, w_lpips)
    return total, named


def stage2_losses(out: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor], w_mse: float = 0.2,
                  w_flow: float = 0.01, w_cycle: float = 0.01, lpips=None, w_lpips: float = 1.0):
    """2nd_State_Conditional_Human-Object/src/model/mipnerf360/model.py:918-944 `get_loss` without the LPIPS term:
    0.2 * MSE on the unpacked patches + 0.01 * flow (weighted by the network's own composite `weights`, all rays)
    + 0.01 * cycle.  `batch` carries `target_rgbs` / `mse_const` / `mse_count` from `prepare_patch_targets`."""
    flow = "deform_pts_prev_final" in out and "ray_grid" in batch
    total, parts = ops.train_losses(
        out["rgb"], batch["target_rgbs"], batch.get("mse_const", 0.0), batch.get("mse_count"),
        pts_prev=out["deform_pts_prev_final"] if flow else None, weights=out["weights"] if flow else None,
        ray_grid=batch.get("ray_grid"), fg=None, cam_prev=batch.get("newsmpl_to_camera_prev"),
        intrinsics_prev=batch.get("intrinsics_prev"), observe=out["observe_pts"], deform=out["deform_pts_final"],
        n_cyc_dev=out.get("cycle_count"), w_mse=w_mse, w_flow=w_flow, w_cycle=w_cycle)
    named = _loss_parts(parts)
    if lpips is not None:
        term = _lpips_term(lpips, out["rgb"], batch, w_lpips)
        total = total + term
        named["lpips"] = _lpips_named(term, w_lpips)
    return total, named


def check_range(modules, device) -> bool:
    """Training-loop side of the fp16 range guard (ops.guarded_forward is the no-grad side): one or two 4-byte reads, so every few
    hundred steps, not every step.  True -- and the given modules switched to exact fp32 MFMA -- if a hidden activation left the
    exactly-representable fp16 hi/lo range since the last poll: either the flag is still set (a forward whose optimiser step has
    not run yet) or the optimiser kernel skipped steps meanwhile (it skips exactly the offending steps and re-arms the word itself,
    so a loop that never polls loses those steps, not the rest of the run)."""
    ops.arm_range_flag(device)
    pending = ops.range_events(device)
    skipped = ops.range_skips(device, since_last_poll=True)
    if not pending and skipped <= 0:
        return False
    for m in modules:
        m.gemm_mode = ops.GEMM_FP32
    return True


def human_lr_decay(step: int, lrate_decay: int = 500) -> float:
    """`optimizer_step` of the human stages (M2:606-634, M:1631-1656): every group's lr = base * 0.1 ** (step / (lrate_decay * 1000))."""
    return 0.1 ** (step / (lrate_decay * 1000.0))


def backward_human(net, loss: torch.Tensor, opt: "FusedAdam", group=None):
    """loss.backward() for a step that contains the human network, data-parallel aware: the volume decoder's backward runs
    after the all-reduce of its 3.5 MB output gradient (Network.decoder_backward), the rest of the flat gradient is
    exchanged here, and `opt.step(..., reduced=True)` must follow.  Equivalent to loss.backward() + a whole-buffer all-reduce."""
    if not net.split_decoder_backward:
        raise RuntimeError("backward_human: set net.split_decoder_backward = True before the forward pass")
    loss.backward()
    finish_backward_human(net, opt, group)


def finish_backward_human(net, opt: Optional["FusedAdam"] = None, group=None):
    """The part of `backward_human` behind `loss.backward()`: exchange the volume gradient, run the decoder's backward on the
    sum, exchange the other spans, and tell the optimiser that this flat gradient is already reduced."""
    net.decoder_backward(group)
    allreduce_flat_grad(net, group, net.reduce_ranges())
    net.split_decoder_backward = False
    if opt is not None:
        opt.grad_is_reduced = True


def train_step_stage2(net, opt: FusedAdam, batch: Dict[str, torch.Tensor], lr: Optional[float] = None, t_rand=None, lpips=None):
    """One stage-2 optimisation step (M2:571-605 training_step + :606-634 optimizer_step): the human-object network with its
    in-network composite, 0.2 MSE on the unpacked patches + 0.01 flow + 0.01 cycle, backward, flat Adam with the
    per-module learning rates.  `batch` comes from `prepare_patch_targets` + `batch_to_device`."""
    opt.zero_grad()
    net.split_decoder_backward = True
    out = net(t_rand=t_rand, static_cycle=True, **batch)
    loss, parts = stage2_losses(out, batch, lpips=lpips)
    backward_human(net, loss, opt, opt.group)
    opt.step(lr, reduced=True)                  # clips by the optimiser's own GradClip (run.grad_max_norm), then Adam
    return loss.detach(), parts


def train_step_stage3(hos, opt_bkgd: FusedAdam, opt_human: FusedAdam, batch: Dict[str, torch.Tensor], lr: Optional[float] = None,
                      jitters=None, t_rand=None, lpips=None):
    """One stage-3 optimisation step (M:1501-1629 + optimizer_step :1631-1656): background forward (3 levels, only
    the NeRF level trains -- the proposal MLPs get no gradient in stage 3) + human branch + merge composite +
    losses + backward + ONE gradient-norm clip over both modules (when the optimisers share a `GradClip`: the Trainer's
    `gradient_clip_val`, 3rd_.../run.py:188-189) + the two flat Adam updates.  `jitters` / `t_rand`: injected sampling draws."""
    opt_bkgd.zero_grad()
    opt_human.zero_grad()
    hos.human.split_decoder_backward = True
    out = hos.render(batch, randomized=True, is_train=True, static_cycle=True, jitters=jitters, t_rand=t_rand)
    loss, parts = stage3_losses(out, batch, lpips=lpips)
    backward_human(hos.human, loss, opt_human, opt_human.group)
    step_all([opt_bkgd, opt_human], lr, reduced=[False, True])       # ONE gradient norm over both modules, then the two Adams
    return loss.detach(), parts
