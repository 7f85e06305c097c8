"""State-conditional human-object renderer (HumanNeRF-style) on MI355X.

Drop-in mirror of `core/nets/human_nerf/network.py::Network` of the reference (stage 3, N:27-698; `stage=2` selects
the stage-2 file's variant, 2nd_State_Conditional_Human-Object/core/nets/human_nerf/network.py:273-299, 538-556, which
composites its own samples with `_raw2outputs` + background colour and returns `rgb, alpha, depth, weights`): same constructor (`Network(cfg)`), same
`forward(rays, dst_Rs, dst_Ts, cnl_gtfms, motion_weights_priors, dst_posevec, near, far, iter_val, **kwargs)`
signature (swallows arbitrary extra kwargs), same output dict, same state_dict keys
(`mweight_vol_decoder.*`, `non_rigid_mlp.*`, `non_rigid_forward_mlp.*`, `cnl_mlp.*`, `pose_decoder.*`,
`human_stateembeds.*`).

Per-step prologue: pose refiner MLP + Rodrigues + 26-joint kinematic chain + motion bases are two HIP launches
(hos_pose_refine_*, hos_motion_basis_*; P2, P3), the 5-layer ConvTranspose3d volume decoder is GEMM + gather kernels
(hos_deconv3d_*; P4).  Everything per sample point (P5-P10) is HIP: sample+backward-LBS warp, hann/Fourier embedders,
split-MFMA MLPs, forward LBS.
"""
from __future__ import annotations

import json
import math
import os
from types import SimpleNamespace
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .flat import FlatModule, FlatStore, Region
from .mipnerf360 import select_state, _Lin

SMPL_PARENT = {1: 0, 2: 0, 3: 0, 4: 1, 5: 2, 6: 3, 7: 4, 8: 5, 9: 6, 10: 7, 11: 8, 12: 9, 13: 9, 14: 9, 15: 12,
               16: 13, 17: 14, 18: 16, 19: 17, 20: 18, 21: 19, 22: 20, 23: 21, 24: 23, 25: 22}   # U:100-103


class Cfg(dict):
    """Minimal attribute-access config (the reference uses a vendored yacs CfgNode; any object with the same
    attributes works, including that CfgNode)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v


def default_cfg(basedir: Optional[str] = None) -> Cfg:
    """configs/default.yaml merged with configs/human_nerf/wild/monocular/adventure.yaml (hot-path fields)."""
    return Cfg(
        basedir=basedir, total_bones=26, N_samples=128, perturb=1.0, chunk=8192, chunk_bkg=8192, netchunk_per_gpu=10000,
        ignore_non_rigid_motions=False,
        canonical_mlp=Cfg(mlp_depth=8, mlp_width=256, multires=10, i_embed=0),
        mweight_volume=Cfg(embedding_size=256, volume_size=32),
        non_rigid_motion_mlp=Cfg(condition_code_size=75, mlp_width=128, mlp_depth=6, skips=[4], multires=6, i_embed=0,
                                 kick_in_iter=100000, full_band_iter=200000),
        non_rigid_forward_mlp=Cfg(condition_code_size=75, mlp_width=128, mlp_depth=6, skips=[4], multires=6, i_embed=0),
        pose_decoder=Cfg(embedding_size=75, mlp_width=256, mlp_depth=4, kick_in_iter=20000),
        # data side (configs/default.yaml:121-146, adventure.yaml:41-42): read by the launcher's scene / evaluation modes
        patch=Cfg(sample_subject_ratio=0.8, N_patches=2, size=32), freeview=Cfg(frame_idx=119), bbox_offset=0.6,
        bgcolor=[255.0, 255.0, 255.0], resize_img_scale=1.0,
    )


def _xavier_(w: torch.Tensor, gain: float, fan_sum: float):
    b = gain * math.sqrt(2.0 / fan_sum) * math.sqrt(3.0)
    w.uniform_(-b, b)


class _LayerSpec:
    __slots__ = ("W", "b", "N", "K", "Kpad", "Npad")

    def __init__(self, store: FlatStore, N: int, K: int, Kpad: Optional[int] = None):
        self.N, self.K = N, K
        self.Kpad, self.Npad = (ops.round_up(K, 32) if Kpad is None else Kpad), ops.round_up(N, 32)
        self.W = store.alloc(N, K, self.Npad, self.Kpad)
        self.b = store.alloc(1, N, 1, self.Npad)

    def bind(self, store: FlatStore) -> _Lin:
        return _Lin(store.bind(self.W, (slice(0, self.N), slice(0, self.K)), (self.N, self.K)),
                    store.bind(self.b, (0, slice(0, self.N)), (self.N,)))


class _Holder(nn.Module):
    pass


NR_LDE = 128      # [cond(75) | hann features(36) | 0]  -> first-layer input row of the non-rigid MLPs
NR_LDPE = 64      # hann features alone (36, zero padded) for the skip concat
CNL_LDE = 128     # [fourier(63) | state(64) | 0]
CNL_CAT = 384     # skip-concat row [fourier+state (127) | h (256) | 0]
CNL_NF = 63       # Fourier features of a canonical point (3 + 6 x 10)
CNL_NFP = 64      # ... padded: the folded forms' input rows [fourier | 0] and concat rows [fourier | 0 | h (256)]


def _round64(n: int) -> int:
    return (n + 63) // 64 * 64


class Network(FlatModule):
    COMPACT_FIRST_DECONV = os.environ.get("HOS_COMPACT_DECONV", "1") != "0"
    gemm_mode = None      # None: the process default (ops.set_gemm_mode); ops.GEMM_* pins the arithmetic of this module's GEMMs

    def __init__(self, cfg, stage: int = 3):
        super().__init__()
        self.cfg = cfg
        self.stage = stage
        K = cfg.total_bones
        if K != 26 or cfg.N_samples > 256 or cfg.canonical_mlp.mlp_width != 256 or cfg.canonical_mlp.mlp_depth != 8 \
                or cfg.canonical_mlp.multires != 10 or cfg.non_rigid_motion_mlp.mlp_width != 128 \
                or cfg.non_rigid_motion_mlp.mlp_depth != 6 or list(cfg.non_rigid_motion_mlp.skips) != [4] \
                or cfg.non_rigid_motion_mlp.multires != 6 or cfg.non_rigid_motion_mlp.condition_code_size != 75 \
                or cfg.mweight_volume.volume_size != 32 or cfg.pose_decoder.mlp_depth != 4 \
                or getattr(cfg, "ignore_non_rigid_motions", False):
            raise NotImplementedError("hosnerf_amd implements the configuration the reference ships (default.yaml + adventure.yaml)")
        st = self.store
        V = cfg.mweight_volume.volume_size

        # ---- parameters that feed torch ops (prologue): natural shapes, still inside the flat buffer
        def plain(shape):
            n = int(np.prod(shape))
            r = st.alloc(1, n)
            return r, tuple(shape)

        pl = {}
        pl["mweight_vol_decoder.const_embedding"] = plain((cfg.mweight_volume.embedding_size,))
        pl["mweight_vol_decoder.decoder.block_mlp.0.weight"] = plain((1024, cfg.mweight_volume.embedding_size))
        pl["mweight_vol_decoder.decoder.block_mlp.0.bias"] = plain((1024,))
        chans, ci, co = [], 1024, 512
        for _ in range(int(np.log2(V)) - 1):                      # U:31-44
            chans.append((ci, co))
            if ci == co:
                co = ci // 2
            else:
                ci = co
        chans.append((ci, K + 1))
        self._deconv_chans = chans
        self._w0c = None
        for n, (a, b) in enumerate(chans):
            pl[f"mweight_vol_decoder.decoder.block_conv.{2 * n}.weight"] = plain((a, b, 4, 4, 4))
            if n == 0 and self.COMPACT_FIRST_DECONV:
                # The first ConvTranspose3d(4, 2, 1) sees ONE input voxel and produces 2^3: output voxel o = 2 * 0 + k - 1 only
                # exists for k in {1, 2} per axis, so 8 of the kernel's 64 taps can ever contribute -- 7/8 of this layer's
                # 33.5 M weights (29.4 M of the network's 64.7 M parameters) never influence the output and never receive a
                # gradient (in the reference neither: Adam leaves them at their initial values).  The reference-shaped parameter
                # stays in the flat buffer for `state_dict` compatibility but is INACTIVE (no kernel, no zeroing, no norm, no
                # Adam touches it); the live taps are this compact [Cin, 8 * Cout] copy (tap-major, so x @ Wc IS the
                # channel-last 2^3 x Cout output), synchronised with the parameter on load / save (`_compact_*`).
                full = pl[f"mweight_vol_decoder.decoder.block_conv.{2 * n}.weight"][0]
                st.inactive.append((full.offset, _round64(full.numel)))
                self._w0c = st.alloc(a, 8 * b)
            pl[f"mweight_vol_decoder.decoder.block_conv.{2 * n}.bias"] = plain((b,))
        pw, pe = cfg.pose_decoder.mlp_width, cfg.pose_decoder.embedding_size
        for name, shp in (("block_mlps.0", (pw, pe)), ("block_mlps.2", (pw, pw)), ("block_mlps.4", (pw, pw)),
                          ("block_mlps_dstR.0", (pw, pw)), ("block_mlps_dstR.2", (3 * (K - 1), pw)),
                          ("block_mlps_dstT.0", (pw, pw)), ("block_mlps_dstT.2", (3 * (K - 1), pw))):
            pl[f"pose_decoder.{name}.weight"] = plain(shp)
            pl[f"pose_decoder.{name}.bias"] = plain((shp[0],))

        # ---- HIP MLPs: zero-padded GEMM layouts
        def nonrigid_specs():
            L = [_LayerSpec(st, 128, 111, NR_LDE)]
            for i in range(1, 6):
                L.append(_LayerSpec(st, 128, 164, 128 + NR_LDPE) if i == 4 else _LayerSpec(st, 128, 128))
            L.append(_LayerSpec(st, 3, 128))
            return L

        self._nr = nonrigid_specs()
        self._nrf = nonrigid_specs()
        self._cnl = [_LayerSpec(st, 256, 127, CNL_LDE)]
        for i in range(1, 8):
            self._cnl.append(_LayerSpec(st, 256, 383, CNL_CAT) if i == 5 else _LayerSpec(st, 256, 256))
        self._cnl.append(_LayerSpec(st, 4, 256))

        tt_path = os.path.join(cfg.basedir, "transitions_times.json") if getattr(cfg, "basedir", None) else None
        if tt_path is not None and os.path.exists(tt_path):
            with open(tt_path, "r") as f:
                infos = json.load(f)
            self.transitions_times = np.stack([np.array(infos[k]["time"], dtype=np.float32) for k in infos], axis=0)
            n_states = self.transitions_times.shape[0] + 1
        else:
            self.transitions_times = None
            n_states = 1
        self._embeds = st.alloc(n_states, 64)
        st.materialize()

        # ---- module tree with the reference's names
        def tree_set(root: nn.Module, dotted: str, param: nn.Parameter):
            parts = dotted.split(".")
            m = root
            for p_ in parts[:-1]:
                if not hasattr(m, p_):
                    setattr(m, p_, _Holder())
                m = getattr(m, p_)
            setattr(m, parts[-1], param)

        self._plain: Dict[str, nn.Parameter] = {}
        for name, (region, shape) in pl.items():
            p_ = st.bind(region, (0, slice(0, region.cols)), shape)
            self._plain[name] = p_
            tree_set(self, name, p_)

        def attach(prefix: str, specs: List[_LayerSpec], attr: str, idxs: List[int]):
            holder = _Holder()
            md = _Holder()
            for L, i in zip(specs, idxs):
                setattr(md, str(i), L.bind(st))
            setattr(holder, attr, md)
            return holder

        self.non_rigid_mlp = attach("non_rigid_mlp", self._nr, "block_mlps", [0, 2, 4, 6, 8, 10, 12])
        self.non_rigid_forward_mlp = attach("non_rigid_forward_mlp", self._nrf, "block_mlps", [0, 2, 4, 6, 8, 10, 12])
        self.cnl_mlp = attach("cnl_mlp", self._cnl[:-1], "pts_linears", [0, 2, 4, 6, 8, 10, 12, 14])
        out_holder = _Holder()
        setattr(out_holder, "0", self._cnl[-1].bind(st))
        self.cnl_mlp.output_linear = out_holder
        self.human_stateembeds = nn.ParameterList([st.bind(self._embeds, (k, slice(None)), (64,)) for k in range(n_states)])
        self.reset_parameters()
        self._token = torch.zeros(1, requires_grad=True)
        self._chain_bufs = {}
        if self._w0c is not None:
            self._compact_from_full()
            self.register_load_state_dict_post_hook(lambda module, incompatible: module._compact_from_full())
            self.register_state_dict_pre_hook(lambda module, prefix, keep_vars: module._compact_to_full())

    # ------------------------------------------------------------------ init (U:181-308 initseq rules)
    @torch.no_grad()
    def reset_parameters(self):
        relu = math.sqrt(2.0)
        lrelu = math.sqrt(2.0 / (1 + 0.2**2))
        P = self._plain
        P["mweight_vol_decoder.const_embedding"].normal_()
        w = P["mweight_vol_decoder.decoder.block_mlp.0.weight"]
        _xavier_(w, lrelu, w.shape[0] + w.shape[1])
        P["mweight_vol_decoder.decoder.block_mlp.0.bias"].zero_()
        for n, (a, b) in enumerate(self._deconv_chans):
            w = P[f"mweight_vol_decoder.decoder.block_conv.{2 * n}.weight"]
            gain = lrelu if n < len(self._deconv_chans) - 1 else 1.0
            _xavier_(w, gain, (a + b) * 8.0)
            base = w[:, :, 0::2, 0::2, 0::2].clone()
            for i in range(2):
                for j in range(2):
                    for k in range(2):
                        w[:, :, i::2, j::2, k::2] = base
            P[f"mweight_vol_decoder.decoder.block_conv.{2 * n}.bias"].zero_()
        for name in ("block_mlps.0", "block_mlps.2", "block_mlps.4", "block_mlps_dstR.0", "block_mlps_dstT.0"):
            w = P[f"pose_decoder.{name}.weight"]
            _xavier_(w, relu, w.shape[0] + w.shape[1])
            P[f"pose_decoder.{name}.bias"].zero_()
        for head in ("dstR", "dstT"):
            P[f"pose_decoder.block_mlps_{head}.2.weight"].uniform_(-1e-5, 1e-5)
            P[f"pose_decoder.block_mlps_{head}.2.bias"].zero_()
        for holder in (self.non_rigid_mlp, self.non_rigid_forward_mlp):
            for i in (0, 2, 4, 6, 8, 10):
                lin = getattr(holder.block_mlps, str(i))
                _xavier_(lin.weight, relu, lin.weight.shape[0] + lin.weight.shape[1])
                lin.bias.zero_()
            last = getattr(holder.block_mlps, "12")
            last.weight.uniform_(-1e-5, 1e-5)
            last.bias.zero_()
        for i in range(0, 16, 2):
            lin = getattr(self.cnl_mlp.pts_linears, str(i))
            _xavier_(lin.weight, relu, lin.weight.shape[0] + lin.weight.shape[1])
            lin.bias.zero_()
        lin = getattr(self.cnl_mlp.output_linear, "0")
        _xavier_(lin.weight, 1.0, lin.weight.shape[0] + lin.weight.shape[1])
        lin.bias.zero_()
        for e in self.human_stateembeds:
            e.normal_()
        if getattr(self, "_w0c", None) is not None and getattr(self.store, "param", None) is not None:
            self._compact_from_full()          # the live taps follow the re-initialised parameter

    # ------------------------------------------------------------------ compact first deconvolution layer
    _LIVE_TAPS = [(od + 1) * 16 + (oh + 1) * 4 + (ow + 1) for od in range(2) for oh in range(2) for ow in range(2)]

    def _first_deconv(self, grad: bool = False):
        """(full [Cin, Cout, 64] view of the reference-shaped parameter, compact [Cin, 8, Cout] view) of weights or gradients."""
        flat = self.store.grad if grad else self.store.param
        fullp = self._plain["mweight_vol_decoder.decoder.block_conv.0.weight"]
        full = (fullp.grad if grad else fullp.detach())
        a, b = fullp.shape[0], fullp.shape[1]
        return full.view(a, b, 64), self._w0c.view(flat).view(a, 8, b)

    @torch.no_grad()
    def _compact_from_full(self):
        """Parameter -> live copy (after `load_state_dict`, `reset_parameters` or any direct write to the parameter)."""
        if self._w0c is None:
            return
        full, comp = self._first_deconv()
        idx = torch.tensor(self._LIVE_TAPS, device=full.device)
        comp.copy_(full.index_select(2, idx).permute(0, 2, 1))

    @torch.no_grad()
    def _compact_to_full(self, grads: bool = False):
        """Live copy -> parameter (before `state_dict()`; `grads=True`: also the gradient, for code that reads `p.grad` of
        every named parameter -- `scatter_compact_grads`)."""
        if self._w0c is None:
            return
        for g in ((False, True) if grads else (False,)):
            full, comp = self._first_deconv(grad=g)
            idx = torch.tensor(self._LIVE_TAPS, device=full.device)
            if g:
                full.zero_()
            full.index_copy_(2, idx, comp.permute(0, 2, 1).contiguous())

    def scatter_compact_grads(self):
        """Make `p.grad` of `mweight_vol_decoder.decoder.block_conv.0.weight` reflect the live taps' gradient (it is not kept
        current during training: nothing in a step reads it).  For inspection / tests that walk `named_parameters()`."""
        self._compact_to_full(grads=True)

    def _after_flat_move(self):
        self._token = torch.zeros(1, device=self.store.param.device, requires_grad=True)

    def _w(self, L: _LayerSpec, grad: bool = False):
        flat = self.store.grad if grad else self.store.param
        return L.W.view(flat), L.b.view(flat).view(-1)

    # ------------------------------------------------------------------ prologue (once per call)
    # Both frames of a training step (current pose and previous-frame pose) go through the prologue as ONE batch of F = 2:
    # pose refinement + Rodrigues + composition is one launch (hos_pose_refine_fwd), the kinematic chain + the two families
    # of motion bases another (hos_motion_basis_fwd); their backward passes are one launch each and write the pose decoder's
    # parameter gradients straight into the flat gradient buffer.  The reference spends ~120 tiny torch launches (and a
    # torch.inverse, which synchronises) on these 26 joints.
    def _pose_tensors(self, grad: bool = False):
        P = self._plain
        out = []
        for name in ops.POSE_PARAM_ORDER:
            for kind in ("weight", "bias"):
                p_ = P[f"pose_decoder.{name}.{kind}"]
                out.append(p_.grad if grad else p_.detach())
        return out

    def _pose_refine(self, Rs, Ts, posevec):
        """N:589-605 + pose_decoders/mlp_delta_body_pose.py + U:66-92.  Rs [F,K,3,3], Ts [F,K,3], posevec [F,75]."""
        self.store.ensure_bound()
        return ops.pose_refine(self._token, Rs, Ts, posevec, self._pose_tensors(False), self._pose_tensors(True))

    def _motion_basis(self, dst_Rs, dst_Ts, cnl_gtfms):
        """U:134-174 for F frames at once: dst_Rs [F,K,3,3], dst_Ts [F,K,3] -> (R_bwd, T_bwd, R_fwd, T_fwd), each [F,K,...]."""
        return ops.motion_basis(dst_Rs, dst_Ts, cnl_gtfms)

    def _motion_weight_volume(self, priors):
        """deconv_vol_decoder.py:34-42 + U:21-59 -> [K+1, V, V, V]."""
        P = self._plain
        self.store.ensure_bound()
        h = ops.decoder_head(P["mweight_vol_decoder.const_embedding"], P["mweight_vol_decoder.decoder.block_mlp.0.weight"],
                             P["mweight_vol_decoder.decoder.block_mlp.0.bias"])                     # [1, 1024] = 1 voxel, channel-last
        n_conv = len(self._deconv_chans)
        comm, sharded = self.decoder_shard, self._shard_layers
        D = 1
        for n in range(n_conv):                  # ConvTranspose3d(4, 2, 1) as GEMM + gather, channel-last (hos_deconv.hip)
            bias = P[f"mweight_vol_decoder.decoder.block_conv.{2 * n}.bias"]
            if n == 0 and self._w0c is not None:
                wc, gwc = self._w0c.view(self.store.param), self._w0c.view(self.store.grad)
                if comm is not None and 0 in sharded:
                    h = ops.deconv3d_first_sharded(h, wc, gwc, bias, n < n_conv - 1, comm)
                else:
                    h = ops.deconv3d_first(h, wc, gwc, bias, n < n_conv - 1)
            elif comm is not None and n in sharded:
                h = ops.deconv3d_sharded(h, P[f"mweight_vol_decoder.decoder.block_conv.{2 * n}.weight"], bias, D, n < n_conv - 1, comm)
            else:
                h = ops.deconv3d(h, P[f"mweight_vol_decoder.decoder.block_conv.{2 * n}.weight"], bias, D, n < n_conv - 1)
            D *= 2
        return ops.volume_softmax(h, priors)     # [V^3, K+1] channel-last logits -> [K+1, V, V, V]

    # ------------------------------------------------------------------ volume decoder sharded over the data-parallel ranks (round 5)
    # The reference replicates the decoder on every DDP rank (deconv_vol_decoder.py:17-42 under run.py:173-190).  Its work does
    # not depend on the rays, so at a fixed GLOBAL batch it is the part of a step that does not shrink with the number of GPUs:
    # ~1.1 ms of the 6 ms 512-ray step (136 MB of weights streamed forward, twice more backward, 253 MB of Adam state).  With
    # `shard_decoder(comm)` rank r owns input-channel rows [r Cin/W, (r+1) Cin/W) of the first `layers` transposed convolutions
    # (25.2 M of the decoder's 34 M live parameters for layers 0-2): it multiplies, differentiates and updates only those; the
    # rows of the other ranks become inactive spans of its flat store (not zeroed, not in its norm, not touched by its Adam) and
    # go stale -- `gather_decoder_shards()` refreshes them from their owners before `state_dict()` / a checkpoint.
    decoder_shard = None            # train.ShardComm
    _shard_layers = ()

    def _shard_rows(self, n: int):
        """(first flat offset, floats per input-channel row, Cin) of sharded layer n's weight."""
        if n == 0 and self._w0c is not None:
            a, b = self._deconv_chans[0]
            return self._w0c.offset, 8 * b, a
        w = self._plain[f"mweight_vol_decoder.decoder.block_conv.{2 * n}.weight"]
        return (w.data_ptr() - self.flat_param.data_ptr()) // 4, w.shape[1] * 64, w.shape[0]

    def shard_decoder(self, comm, layers=(0, 1, 2)):
        """Call once, after `.to(device)` and BEFORE the optimiser is built (its spans are cut from the store's active spans)."""
        if self.decoder_shard is not None:
            raise RuntimeError("the volume decoder is already sharded")
        if comm.world == 1:
            return self
        for n in layers:
            off, row, cin = self._shard_rows(n)
            if cin % (comm.world * 32):
                raise ValueError(f"decoder layer {n}: {cin} input channels do not split into {comm.world} shards of a multiple of 32")
            cs = cin // comm.world
            lo, hi = off + comm.rank * cs * row, off + (comm.rank + 1) * cs * row
            if lo > off:
                self.store.inactive.append((off, lo - off))
            if hi < off + cin * row:
                self.store.inactive.append((hi, off + cin * row - hi))
        self.decoder_shard, self._shard_layers = comm, tuple(layers)
        return self

    def decoder_shard_spans(self):
        """[(offset, numel)] of the rows THIS rank owns (their gradient exists on this rank only: `train._shard_norm_correction`)."""
        out = []
        if self.decoder_shard is not None:
            for n in self._shard_layers:
                off, row, cin = self._shard_rows(n)
                cs = cin // self.decoder_shard.world
                out.append((off + self.decoder_shard.rank * cs * row, cs * row))
        return out

    @torch.no_grad()
    def gather_decoder_shards(self, optimizer=None):
        """Collective (every rank calls it): each sharded layer's rows are fetched from their owners, so that the flat parameter
        buffer -- and with it `state_dict()` -- is complete and identical on every rank again.  `optimizer` (the FusedAdam of this
        module, or an object with a `.fused` list that contains it): its two moment buffers -- same flat layout -- are completed the
        same way, so that the optimiser state rank 0 writes into a checkpoint is the state of ALL rows and a resumed run continues
        every shard's Adam where it stopped (without it the rows of ranks >= 1 restart with m = v = 0 at step t)."""
        comm = self.decoder_shard
        if comm is None:
            return
        flats = [self.flat_param]
        for f in (getattr(optimizer, "fused", None) or ([optimizer] if optimizer is not None else [])):
            if getattr(f, "module", None) is self:
                flats += [f.exp_avg, f.exp_avg_sq]
        for n in self._shard_layers:
            off, row, cin = self._shard_rows(n)
            cs = cin // comm.world
            for flat in flats:
                mine = flat[off + comm.rank * cs * row: off + (comm.rank + 1) * cs * row]
                flat[off: off + cin * row].copy_(comm.all_gather(mine).reshape(-1))

    # ------------------------------------------------------------------ data-parallel backward of the volume decoder
    # The motion-weight volume decoder (63.4 M of the 64.7 M parameters, 253 MB of gradient) sees no ray: its input is a learned
    # constant, so its forward is IDENTICAL on every rank and only the gradient that arrives at its 3.5 MB output differs.
    # Backpropagation is linear, so summing that output gradient over the ranks FIRST and running the decoder's backward on
    # the sum gives every rank the already-reduced parameter gradients: the per-step exchange of this module shrinks from
    # 259 MB to 3.5 MB (volume gradient) + 6 MB (every other parameter) -- what makes strong scaling over xGMI possible.
    split_decoder_backward = False
    _pending_vol = None
    _fwd_stream = None          # the stream the last training forward was queued on

    def lazy_param_spans(self):
        """[(offset, numel)] of the flat buffer that can be WITHOUT a gradient in a training step (torch's Adam skips such parameters):
        every state embedding (one state per call, N:179-246) and the pose decoder, which is not evaluated before
        `cfg.pose_decoder.kick_in_iter` (N:589-605; all its parameters start together, so one span = one step count)."""
        spans = [(self._embeds.offset + k * 64, 64) for k in range(len(self.human_stateembeds))]
        first = self._plain["pose_decoder.block_mlps.0.weight"]
        lo = (first.data_ptr() - self.flat_param.data_ptr()) // 4
        spans.append((int(lo), int(self._nr[0].W.offset - lo)))
        return spans

    def decoder_span(self):
        """(offset, numel) of the volume decoder's parameters in the flat buffer (they are allocated first)."""
        first = self._plain["pose_decoder.block_mlps.0.weight"]
        off = (first.data_ptr() - self.flat_param.data_ptr()) // 4
        return 0, int(off)

    def pending_volume_grad(self):
        """The gradient w.r.t. the volume left by the first half of a split backward (None if the graph was not cut) -- the
        3.5 MB tensor a data-parallel step all-reduces (sum) before `finish_decoder_backward`."""
        return None if self._pending_vol is None else self._pending_vol[1].grad

    def finish_decoder_backward(self):
        """Backpropagate the (reduced) volume gradient through the decoder: its parameter gradients land in the flat buffer."""
        pend, self._pending_vol = self._pending_vol, None
        if pend is not None and pend[1].grad is not None:
            pend[0].backward(pend[1].grad)
            # the decoder's forward may have run on a side stream (HOSNeRF.render): autograd runs this backward there, and its HIP
            # weight-gradient kernels write the flat gradient directly -- the calling stream must wait for them
            fs = self._fwd_stream
            if fs is not None and pend[0].is_cuda and fs != torch.cuda.current_stream(pend[0].device):
                torch.cuda.current_stream(pend[0].device).wait_stream(fs)

    def decoder_backward(self, group=None):
        """Second half of a split backward (`split_decoder_backward = True`): all-reduce (sum) the gradient w.r.t. the volume,
        then backpropagate it through the decoder.  No-op if the last forward did not cut the graph."""
        g = self.pending_volume_grad()
        if g is not None:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
                dist.all_reduce(g, group=group)
        self.finish_decoder_backward()

    def reduce_ranges(self):
        """Spans of the flat gradient that still need the data-parallel all-reduce when the decoder's backward is split."""
        off, n = self.decoder_span()
        return [(off + n, self.flat_param.numel() - off - n)]

    def _band_weights(self, iter_val: float, device) -> torch.Tensor:
        """hannw_fourier.py:29-44 (evaluated with torch on the host, 6 floats).  The device copy is cached by value: before
        kick-in and after full band the weights do not change, and a captured step must not issue a host->device copy."""
        c = self.cfg.non_rigid_motion_mlp
        kick = torch.tensor(float(c.kick_in_iter), dtype=torch.float32)
        t = torch.clamp(torch.tensor(float(iter_val), dtype=torch.float32) - kick, min=0.0)
        Nn = c.full_band_iter - kick
        alpha = c.multires * t / Nn
        j = torch.arange(c.multires, dtype=torch.float32)
        w = (1.0 - torch.cos(np.pi * torch.clamp(alpha - j, min=0.0, max=1.0))) / 2.0
        key = (tuple(w.tolist()), str(device))
        cache = getattr(self, "_band_cache", None)
        if cache is None or cache[0] != key:
            self._band_cache = (key, w.to(device))
        return self._band_cache[1]

    # ------------------------------------------------------------------ HIP MLP chains (forward)
    def _nonrigid_fwd(self, specs: List[_LayerSpec], x: torch.Tensor, cond: torch.Tensor, band_w: torch.Tensor, save: bool, rows_dev=None):
        """mlp_offset.py:54-70: xyz + MLP([cond | hann(x)]) with the hann features re-concatenated before Linear #4.
        `rows_dev` (int32 [1] on the device): only that many leading rows are live (fixed-capacity cycle set)."""
        Pn = x.shape[0]
        dev = x.device
        PE = torch.empty(Pn, NR_LDPE, device=dev)
        chain = ops.MLP_CHAIN and Pn >= ops.MLP_CHAIN_MIN_ROWS and ops.get_gemm_mode() != ops.GEMM_FP32
        if chain and ops.MLP_CHAIN_FOLD:
            # The condition code is one vector per FRAME (mlp_offset.py:55 expands it over the points): W0[:, :75] . cond is a
            # bias of this launch, the first layer reduces over the 36 hann features only, and the [P, 128] first-layer rows are
            # never built -- PE serves layer 0 and the skip layer (hos_chain.hip, FOLD).
            cond = cond.reshape(-1).contiguous()
            ops.embed_hannw(x, band_w, None, PE, rows_dev=rows_dev)
            key = (id(specs),) + ops._stream_key(dev)          # per (device, stream): the folded bias belongs to this call's frame
            bufs = self._chain_bufs.get(key)
            if bufs is None or len(bufs) < 3:
                bufs = self._chain_bufs[key] = ops.mlp_chain_buffers(dev) + (torch.empty(128, 64, device=dev),)
            ws = [self._w(L) for L in specs]
            ops.mlp_chain_pack_fold([w for w, _ in ws], [b_ for _, b_ in ws], cond, 6 * band_w.numel(), bufs[0], bufs[1], bufs[2])
            acts = [torch.empty(Pn, 128, device=dev) for _ in range(6)]
            xyz = torch.empty(Pn, 3, device=dev)
            ops.mlp_chain128_fwd(None, PE, x, bufs[0], bufs[1], acts, xyz, rows_dev=rows_dev)
            return xyz, ((None, PE, acts, (bufs[2], cond)) if save else None)
        E = torch.empty(Pn, NR_LDE, device=dev)
        ops.embed_hannw(x, band_w, cond.reshape(-1), E, PE, rows_dev=rows_dev)
        if chain:
            # the whole MLP in one launch, activations on chip across the layers (hos_chain.hip)
            key = id(specs)
            bufs = self._chain_bufs.get(key)
            if bufs is None or bufs[0].device != dev:
                bufs = self._chain_bufs[key] = ops.mlp_chain_buffers(dev)
            ws = [self._w(L) for L in specs]
            ops.mlp_chain_pack([w for w, _ in ws], [b_ for _, b_ in ws], bufs[0], bufs[1])
            acts = [torch.empty(Pn, 128, device=dev) for _ in range(6)]
            xyz = torch.empty(Pn, 3, device=dev)
            ops.mlp_chain128_fwd(E, PE, x, bufs[0], bufs[1], acts, xyz, rows_dev=rows_dev)
            return xyz, ((E, PE, acts, None) if save else None)
        acts = []
        h = E
        for i in range(6):
            L = specs[i]
            Wt, bt = self._w(L)
            out = torch.empty(Pn, 128, device=dev)
            if i == 4:
                ops.linear_fwd(h, 128, Wt, bt, 128, out, ops.EPI_RELU, A1=PE, K1=NR_LDPE, rows_dev=rows_dev)
            else:
                ops.linear_fwd(h, L.Kpad, Wt, bt, 128, out, ops.EPI_RELU, rows_dev=rows_dev)
            acts.append(out)
            h = out
        Wt, bt = self._w(specs[6])
        xyz = torch.empty(Pn, 3, device=dev)
        ops.linear_fwd(h, 128, Wt, bt, 3, xyz, ops.EPI_RESIDUAL, aux=x, rows_dev=rows_dev)
        return xyz, ((E, PE, acts, None) if save else None)

    def _canonical_fwd(self, cnl: torch.Tensor, state: int, save: bool):
        """mlp_rgb_sigma.py:49-58 + N:539-540: [P,4] = (sigmoid rgb, relu sigma)."""
        Pn = cnl.shape[0]
        dev = cnl.device
        embed = self._embeds.view(self.store.param)[state]
        # The state embedding is ONE vector per call (N:177-230 picks it by frame time): its 64 columns of the input layer and of
        # the skip layer are biases of this call.  Folded form (hos_thin.hip: hos_canonical_fold_*): input rows [fourier 63 | 0],
        # concat rows [fourier 63 | 0 | h 256] with the h part 16-byte aligned -- 64 / 320 columns instead of 128 / 384, and the
        # 320-wide skip layer fits the register-resident thin kernel (20 reduction steps) instead of the tiled GEMM.
        fold = None
        if ops.CNL_FOLD and ops.thin_dgrad_rows(Pn):
            fkey = ("cnl_fold",) + ops._stream_key(dev)       # per (device, stream): packed with this call's embedding, read by its backward
            fb = self._chain_bufs.get(fkey)
            if fb is None:
                fb = self._chain_bufs[fkey] = torch.empty(256 * (2 * CNL_NFP + 256 + 2), device=dev)
            fw = ops.cnl_fold_views(fb, 256, CNL_NFP, 256)
            (W0, b0), (W5, b5) = self._w(self._cnl[0]), self._w(self._cnl[5])
            ops.canonical_fold_pack(W0, b0, W5, b5, embed, 256, CNL_NF, 256, fw)
            E = torch.empty(Pn, CNL_NFP, device=dev)
            CAT = torch.empty(Pn, CNL_NFP + 256, device=dev)
            ops.embed_fourier(cnl, 10, ops.zero1(dev), E, CAT)          # a one-element zero "state" = the pad column of both rows
            fold = (fw, embed)
        else:
            E = torch.empty(Pn, CNL_LDE, device=dev)
            CAT = torch.empty(Pn, CNL_CAT, device=dev)
            CAT[:, CNL_CAT - 1].zero_()
            ops.embed_fourier(cnl, 10, embed, E, CAT)
        acts, bits = [], []
        h = E
        # training: every layer on the thin kernel also writes its ReLU mask as one bit per element (1 KB per 32 rows), which the
        # backward reads instead of the fp32 activations (a third of a thin dgrad launch's HBM traffic)
        want_bits = save and ops.RELU_BITS and ops.thin_dgrad_rows(Pn)
        for i in range(8):
            L = self._cnl[i]
            Wt, bt = self._w(L)
            K = L.Kpad
            if fold is not None and i in (0, 5):
                Wt, bt = (fold[0][0], fold[0][1]) if i == 0 else (fold[0][2], fold[0][3])
                K = Wt.shape[1]
            rb = ops.thin_relu_bits(Pn, dev) if (want_bits and K <= 320) else None
            if i == 4:      # its output feeds the skip concat: write it behind the input columns of CAT
                ops.linear_fwd(h, K, Wt, bt, 256, CAT, ops.EPI_RELU, out_col0=CNL_NFP if fold is not None else 127, relu_bits=rb)
                acts.append(CAT)
                h = CAT
            else:
                out = torch.empty(Pn, 256, device=dev)
                ops.linear_fwd(h, K, Wt, bt, 256, out, ops.EPI_RELU, relu_bits=rb)
                acts.append(out)
                h = out
            bits.append(rb)
        Wt, bt = self._w(self._cnl[8])
        raw = torch.empty(Pn, 4, device=dev)
        ops.linear_fwd(h, 256, Wt, bt, 4, raw, ops.EPI_SIGMOID_RELU4)
        return raw, ((E, acts, bits, fold) if save else None)

    # ------------------------------------------------------------------ HIP MLP chains (backward)
    def _nonrigid_bwd(self, specs: List[_LayerSpec], saved, x: torch.Tensor, band_w: torch.Tensor, g_xyz: torch.Tensor, rows_dev=None):
        """Parameter gradients into the flat buffer; returns d loss / d x  ([P,3]).  Every layer is 128 wide, so each
        layer's (wgrad, dgrad) pair is one fused pass over (dZ, X) (ops.linear_bwd_fused); HOS_FUSED_BWD=0 selects the
        two-GEMM form."""
        E, PE, acts, fold = saved
        Pn, dev = x.shape[0], x.device
        fused = ops.FUSED_THIN_BWD

        def layer_bwd(dz, X, spec, N, K, out, relu_mask, w_col0=0, bias=True, override=None):
            if override is not None:
                Wt, gW, gb = override
            else:
                Wt, _ = self._w(spec)
                gW, gb = self._w(spec, grad=True)
            if fused:
                ops.linear_bwd_fused(dz, X, Wt, gW, gb if bias else None, N, K, out, relu_mask, w_col0=w_col0, rows_dev=rows_dev)
            else:
                ops.linear_wgrad(dz, X, gW, gb if bias else None, N, K, w_col0=w_col0)
                ops.linear_dgrad(dz, Wt, dz.shape[1], K, out, mask_src=X if relu_mask else None, w_col0=w_col0)
            return out

        dz6 = torch.empty(Pn, 32, device=dev)
        ops.slice_pad(g_xyz, 0, 3, dz6, rows_dev=rows_dev)           # [P,3] -> zero-padded [P,32] operand rows, one launch
        if fold is not None:
            # gradient buffers of the folded first layer: [128, 64] for the hann columns of W0 + [128] for the folded bias
            gfold = ops.zero_(ops.fold_grad_workspace(dev))
            gw0h, db0 = gfold[:128 * 64].view(128, 64), gfold[128 * 64:]
        if (fold is not None and fused and ops.MLP_CHAIN_BWD and Pn >= ops.MLP_CHAIN_BWD_MIN_ROWS
                and ops.get_gemm_mode() != ops.GEMM_FP32):
            # Three group launches (hos_mlpbwd.hip, chain_bwd_kernel): the gradient with respect to a layer's output stays in LDS
            # between the layers of a group; only the hand-overs (dz4, dz2), the hann-column gradients and the activations move.
            Wm = [self._w(L)[0] for L in specs]
            G = [self._w(L, grad=True) for L in specs]
            im = ops.mlp_chain_bwd_images(id(specs), (0, 1, 2), dev)
            ops.mlp_chain_bwd_pack([
                (0, 0, Wm[6], 0, 3, 128, im[0][0]), (0, 1, Wm[5], 0, 128, 128, im[0][1]),
                (1, 0, Wm[4], 128, 128, NR_LDPE, im[1][0]), (1, 1, Wm[4], 0, 128, 128, im[1][1]), (1, 2, Wm[3], 0, 128, 128, im[1][2]),
                (2, 0, Wm[2], 0, 128, 128, im[2][0]), (2, 1, Wm[1], 0, 128, 128, im[2][1]), (2, 2, fold[0], 0, 128, NR_LDPE, im[2][2])])
            dPE = torch.empty(Pn, NR_LDPE, device=dev)
            dE = torch.empty(Pn, NR_LDPE, device=dev)
            dz4 = torch.empty(Pn, 128, device=dev)
            dz2 = torch.empty(Pn, 128, device=dev)
            with ops.deferred_bwd_reduce():      # the eight slab reductions as one launch at the end
                ops.mlp_chain_bwd(0, dz6, [acts[5], acts[4]], im[0], [None, dz4], [G[6][0], G[5][0]], [0, 0], [G[6][1], G[5][1]],
                                  [3, 128], [128, 128], rows_dev=rows_dev)
                ops.mlp_chain_bwd(1, dz4, [PE, acts[3], acts[2]], im[1], [dPE, None, dz2], [G[4][0], G[4][0], G[3][0]], [128, 0, 0],
                                  [None, G[4][1], G[3][1]], [128, 128, 128], [NR_LDPE, 128, 128], rows_dev=rows_dev)
                ops.mlp_chain_bwd(2, dz2, [acts[1], acts[0], PE], im[2], [None, None, dE], [G[2][0], G[1][0], gw0h], [0, 0, 0],
                                  [G[2][1], G[1][1], db0], [128, 128, 128], [128, 128, NR_LDPE], rows_dev=rows_dev)
            res = g_xyz.contiguous()
            gW0, gb0 = self._w(specs[0], grad=True)
            ops.mlp_chain_unfold_grad(gw0h, db0, fold[1], 6 * band_w.numel(), gW0, gb0)
            return ops.embed_bwd_res(x, band_w, band_w.numel(), False, dE, 0, dPE, 0, res, rows_dev=rows_dev)
        with ops.deferred_bwd_reduce():          # the seven slab reductions of this chain as one launch at the end
            dz = layer_bwd(dz6, acts[5], specs[6], 3, 128, torch.empty(Pn, 128, device=dev), True)
            dPE = dE = None
            for i in range(5, -1, -1):
                if i == 4:
                    dPE = layer_bwd(dz, PE, specs[4], 128, NR_LDPE, torch.empty(Pn, NR_LDPE, device=dev), False, w_col0=128, bias=False)
                    dz = layer_bwd(dz, acts[3], specs[4], 128, 128, torch.empty(Pn, 128, device=dev), True)
                elif i == 0 and fold is not None:
                    dE = layer_bwd(dz, PE, specs[0], 128, NR_LDPE, torch.empty(Pn, NR_LDPE, device=dev), False, override=(fold[0], gw0h, db0))
                elif i == 0:
                    dE = layer_bwd(dz, E, specs[0], 128, NR_LDE, torch.empty(Pn, NR_LDE, device=dev), False)
                else:
                    dz = layer_bwd(dz, acts[i - 1], specs[i], 128, 128, torch.empty(Pn, 128, device=dev), True)
        # residual path of xyz = x + offset: g_x = g_xyz + d(features)/dx in the embedder's backward launch (no clone)
        res = g_xyz.contiguous()
        if fold is not None:
            gW0, gb0 = self._w(specs[0], grad=True)
            ops.mlp_chain_unfold_grad(gw0h, db0, fold[1], 6 * band_w.numel(), gW0, gb0)
            return ops.embed_bwd_res(x, band_w, band_w.numel(), False, dE, 0, dPE, 0, res, rows_dev=rows_dev)
        return ops.embed_bwd_res(x, band_w, band_w.numel(), False, dE, 75, dPE, 0, res, rows_dev=rows_dev)

    def _canonical_bwd(self, saved, cnl: torch.Tensor, raw: torch.Tensor, g_raw: torch.Tensor, state: int):
        E, acts, bits, fold = saved
        CAT = acts[4]
        Pn, dev = cnl.shape[0], cnl.device
        dz8 = torch.empty(Pn, 32, device=dev)
        ops.rgbsigma_grad(g_raw.contiguous(), raw, dz8)               # writes the whole zero-padded row
        Wt, _ = self._w(self._cnl[8])
        gW, gb = self._w(self._cnl[8], grad=True)
        ops.linear_wgrad(dz8, acts[7], gW, gb, 4, 256)
        dz = torch.empty(Pn, 256, device=dev)
        ops.linear_dgrad(dz8, Wt, 32, 256, dz, mask_src=acts[7], mask_bits=bits[7])
        dCAT = dE = None
        tmp_b = {}
        if fold is not None:
            # gradient buffers of the two folded layers (zeroed: the launches below accumulate), unfolded after the reductions
            gws = ops.zero_(ops.cnl_fold_grad_workspace(dev, 256, CNL_NFP, 256))
            gfold = ops.cnl_fold_views(gws, 256, CNL_NFP, 256)
        with ops.deferred_bwd_reduce():          # the slab reductions of the eight 256-wide weight gradients as one launch at the end
            for i in range(7, -1, -1):
                L = self._cnl[i]
                Wt, _ = self._w(L)
                gW, gb = self._w(L, grad=True)
                if i == 5 and fold is not None:
                    W5f = fold[0][2]
                    ops.linear_wgrad(dz, CAT, gfold[2], gfold[3], 256, CNL_NFP + 256)
                    nxt = torch.empty(Pn, 256, device=dev)
                    dCAT = torch.empty(Pn, CNL_NFP, device=dev)
                    ops.linear_dgrad(dz, W5f, 256, CNL_NFP, dCAT, thin=True)
                    ops.linear_dgrad(dz, W5f, 256, 256, nxt, mask_src=CAT, w_col0=CNL_NFP, mask_col0=CNL_NFP, mask_bits=bits[4])
                    dz = nxt
                elif i == 0 and fold is not None:
                    ops.linear_wgrad(dz, E, gfold[0], gfold[1], 256, CNL_NFP)
                    dE = torch.empty(Pn, CNL_NFP, device=dev)
                    ops.linear_dgrad(dz, fold[0][0], 256, CNL_NFP, dE, thin=True)
                elif i == 5:
                    tmp_b[5] = ops.zeros(L.Npad, dev)
                    ops.linear_wgrad(dz, CAT, gW, tmp_b[5], 256, CNL_CAT)
                    nxt = torch.empty(Pn, 256, device=dev)
                    if ops.thin_dgrad_rows(Pn):
                        # the two consumers of d(concat row) take their own column windows straight from the weight: the
                        # Fourier part (63 columns; the state embedding's gradient goes through db below) and the h part
                        # (columns 127..382, through layer 4's ReLU) -- no [P, 384] round trip, no slice + mask pass
                        dCAT = torch.empty(Pn, 64, device=dev)
                        ops.linear_dgrad(dz, Wt, 256, 64, dCAT, thin=True)
                        ops.linear_dgrad(dz, Wt, 256, 256, nxt, mask_src=CAT, w_col0=127, mask_col0=127, mask_bits=bits[4])
                    else:
                        dCAT = torch.empty(Pn, CNL_CAT, device=dev)
                        ops.linear_dgrad(dz, Wt, 256, CNL_CAT, dCAT)
                        ops.slice_mask(dCAT, 127, CAT, 127, 256, nxt)             # h-part of the concat, through layer 4's ReLU
                    dz = nxt
                elif i == 0:
                    tmp_b[0] = ops.zeros(L.Npad, dev)
                    ops.linear_wgrad(dz, E, gW, tmp_b[0], 256, CNL_LDE)
                    dE = torch.empty(Pn, CNL_LDE, device=dev)
                    ops.linear_dgrad(dz, Wt, 256, CNL_LDE, dE)
                else:
                    inp = acts[i - 1]
                    ops.linear_wgrad(dz, inp, gW, gb, 256, 256)
                    nxt = torch.empty(Pn, 256, device=dev)
                    ops.linear_dgrad(dz, Wt, 256, 256, nxt, mask_src=inp, mask_bits=bits[i - 1])
                    dz = nxt
        g_embed = self._embeds.view(self.store.grad)[state]
        if fold is not None:
            (W0, _), (W5, _) = self._w(self._cnl[0]), self._w(self._cnl[5])
            (gW0, gb0), (gW5, gb5) = self._w(self._cnl[0], grad=True), self._w(self._cnl[5], grad=True)
            ops.canonical_fold_unfold(gfold, W0, W5, fold[1], 256, CNL_NF, 256, gW0, gb0, gW5, gb5, g_embed)
        else:
            # state embedding: its 64 columns are constant over samples -> d embed = db @ W[:, 63:127] (layers 0 and 5)
            for i in (0, 5):
                Wt, _ = self._w(self._cnl[i])
                _, gb = self._w(self._cnl[i], grad=True)
                gb += tmp_b[i]
                g_embed.addmv_(Wt[:256, 63:127].t(), tmp_b[i][:256])
        g_cnl = torch.empty(Pn, 3, device=dev)
        ops.embed_bwd(cnl, None, 10, True, dE, 0, dCAT, 0, g_cnl, False)
        return g_cnl

    # ------------------------------------------------------------------ reference-style forward
    def frame_prologue(self, dst_Rs, dst_Ts, cnl_gtfms, motion_weights_priors, dst_posevec=None, iter_val=1e7, **kwargs):
        """The per-FRAME part of `forward` (N:589-671): pose refinement, motion bases, band weights, pose condition and the
        motion-weight volume.  None of it depends on the rays, so the evaluation loops (which call the reference network
        once per 8192-ray chunk, M:1334-1352, and rebuild the 253 MB deconvolution volume every time) compute it once per
        frame and pass it to `forward(prologue=...)`."""
        cfg = self.cfg
        K = cfg.total_bones
        iter_v = float(torch.as_tensor(iter_val).reshape(-1)[0])
        time = float(kwargs["time"])
        is_train = bool(kwargs["is_train"])
        flow = time > 0.005 and is_train
        nr_kick = cfg.non_rigid_motion_mlp.kick_in_iter

        def cond_of(pv):
            return torch.zeros_like(pv) * pv if iter_v < nr_kick else pv          # N:653-656

        # prologue for the current (and, for the flow set, the previous) frame in one batch
        if flow and dst_Rs.is_cuda and not (dst_Rs.requires_grad or dst_Ts.requires_grad or dst_posevec.requires_grad):
            # the three stacks as ONE launch into one allocation (the inputs are batch tensors: no gradient flows to them)
            nR, nT, nP = dst_Rs.numel(), dst_Ts.numel(), dst_posevec.numel()
            pad4 = lambda n: (2 * n + 3) // 4 * 4
            buf = torch.empty(pad4(nR) + pad4(nT) + pad4(nP), device=dst_Rs.device)
            Rs = buf[:2 * nR].view((2,) + tuple(dst_Rs.shape))
            Ts = buf[pad4(nR):pad4(nR) + 2 * nT].view((2,) + tuple(dst_Ts.shape))
            pv = buf[pad4(nR) + pad4(nT):pad4(nR) + pad4(nT) + 2 * nP].view(2, nP)
            c = lambda t_: t_.detach().float().contiguous()
            ops.copy_or_zero_n([Rs[0], Rs[1], Ts[0], Ts[1], pv[0], pv[1]],
                               [c(dst_Rs), c(kwargs["dst_Rs_prev"]), c(dst_Ts), c(kwargs["dst_Ts_prev"]), c(dst_posevec).reshape(-1),
                                c(kwargs["dst_posevec_prev"]).reshape(-1)])
        elif flow:
            Rs = torch.stack([dst_Rs, kwargs["dst_Rs_prev"]], 0)
            Ts = torch.stack([dst_Ts, kwargs["dst_Ts_prev"]], 0)
            pv = torch.stack([dst_posevec.reshape(-1), kwargs["dst_posevec_prev"].reshape(-1)], 0)
        else:
            Rs, Ts, pv = dst_Rs[None], dst_Ts[None], dst_posevec.reshape(1, -1)
        if iter_v >= cfg.pose_decoder.get("kick_in_iter", 0):
            Rs, Ts = self._pose_refine(Rs, Ts, pv)
        Rb_, Tb_, Rf_, Tf_ = self._motion_basis(Rs, Ts, cnl_gtfms)
        vol = self._motion_weight_volume(motion_weights_priors)
        if self.split_decoder_backward and torch.is_grad_enabled() and vol.requires_grad:
            # data-parallel training: cut the autograd graph at the volume (see decoder_backward)
            self._fwd_stream = torch.cuda.current_stream(vol.device) if vol.is_cuda else None
            leaf = vol.detach().requires_grad_(True)
            self._pending_vol = (vol, leaf)
            vol = leaf
        Rb, Tb, Rf, Tf = ops.unbind_frames_many(Rb_, Tb_, Rf_, Tf_)      # per-frame views; ONE launch stacks all their cotangents
        # the volume feeds the backward warp (channel-major) and, as a channel-last copy of its K bone channels, the forward warp
        vol, vol_cl = ops.volume_pair(vol, K)
        pro = {"flow": flow, "state": select_state(time, self.transitions_times), "R_b": Rb[0], "T_b": Tb[0],
               "R_f": Rf[0], "T_f": Tf[0], "band_w": self._band_weights(iter_v, dst_Rs.device),
               "cond": cond_of(dst_posevec).contiguous(), "vol": vol, "vol_cl": vol_cl}
        if flow:
            pro.update(R_fp=Rf[1], T_fp=Tf[1], cond_prev=cond_of(kwargs["dst_posevec_prev"]).contiguous())
        return pro

    def forward(self, rays, dst_Rs=None, dst_Ts=None, cnl_gtfms=None, motion_weights_priors=None, dst_posevec=None, near=None,
                far=None, iter_val=1e7, t_rand=None, prologue=None, with_cycle: bool = True, static_cycle: bool = False, **kwargs):
        """Reference signature + keyword-only extensions: `t_rand` (the stratified jitter draws), `prologue` (a cached
        `frame_prologue`), `with_cycle=False` (evaluation loops never read the cycle outputs), `static_cycle=True`
        (fixed-shape cycle outputs [P,3] + `cycle_count` on the device instead of the reference's data-dependent
        [n_selected,3]: no host synchronisation, the step can be captured in a hipGraph).
        `self.gemm_mode` (None = process default) selects the arithmetic of this module's GEMMs (see ops.guarded_forward)."""
        return ops.guarded_forward(self, rays.device, lambda: self._forward(
            rays, dst_Rs, dst_Ts, cnl_gtfms, motion_weights_priors, dst_posevec, near, far, iter_val, t_rand, prologue, with_cycle,
            static_cycle, **kwargs))

    def _forward(self, rays, dst_Rs=None, dst_Ts=None, cnl_gtfms=None, motion_weights_priors=None, dst_posevec=None, near=None,
                 far=None, iter_val=1e7, t_rand=None, prologue=None, with_cycle: bool = True, static_cycle: bool = False, **kwargs):
        cfg = self.cfg
        dev = rays.device
        K = cfg.total_bones
        pro = prologue if prologue is not None else self.frame_prologue(
            dst_Rs, dst_Ts, cnl_gtfms, motion_weights_priors, dst_posevec=dst_posevec, iter_val=iter_val, **kwargs)
        flow, state, band_w, cond, vol, vol_cl = pro["flow"], pro["state"], pro["band_w"], pro["cond"], pro["vol"], pro["vol_cl"]
        R_b, T_b, R_f, T_f = pro["R_b"], pro["T_b"], pro["R_f"], pro["T_f"]
        if flow:
            R_fp, T_fp, cond_prev = pro["R_fp"], pro["T_fp"], pro["cond_prev"]
        bmin = kwargs["cnl_bbox_min_xyz"].contiguous()
        bscale = kwargs["cnl_bbox_scale_xyz"].contiguous()

        rays_o = rays[0].reshape(-1, 3).float().contiguous()
        rays_d = rays[1].reshape(-1, 3).float().contiguous()
        B = rays_o.shape[0]
        N = cfg.N_samples
        if t_rand is None and cfg.perturb > 0.0:
            t_rand = torch.rand(B, N, device=dev)                                  # N:421
        outs: Dict[str, List[torch.Tensor]] = {}
        chunk = int(cfg.chunk)
        for c0 in range(0, B, chunk):
            sl = slice(c0, min(B, c0 + chunk))
            tr = None if t_rand is None else t_rand[sl].contiguous()
            grad = torch.is_grad_enabled()
            if grad:
                z, pts, x_skel, mask = ops.human_sample_warp_ad(vol, R_b, T_b, rays_o[sl], rays_d[sl], near[sl].contiguous(),
                                                                far[sl].contiguous(), N, bmin, bscale, tr, K)
                cnl = _NonRigidFn.apply(self._token, self, "nr", x_skel, cond, band_w, None)
                # the canonical points feed up to three consumers (canonical MLP, flow set, cycle set): their cotangents are
                # summed by ONE launch instead of one autograd accumulation per extra consumer
                n_use = 1 + int(flow) + int(with_cycle)
                cnl_uses = list(ops.fanout(cnl, n_use))
                raw = _CanonicalFn.apply(self._token, self, cnl_uses.pop(), state)
            else:
                z, pts, x_skel, mask = ops.human_sample_warp(rays_o[sl], rays_d[sl], near[sl].contiguous(), far[sl].contiguous(),
                                                             N, R_b, T_b, vol, bmin, bscale, tr, K)
                cnl, _ = self._nonrigid_fwd(self._nr, x_skel, cond, band_w, save=False)
                cnl_uses = [cnl, cnl, cnl]
                raw, _ = self._canonical_fwd(cnl, state, save=False)
            b = z.shape[0]
            if self.stage == 2:
                # N2:273-299, 538-556: the stage-2 network composites its own samples (last interval 1e10, masked alphas,
                # background colour added) and returns the maps instead of the per-sample radiance
                rgb_map, acc_map, weights, depth_map = ops.raw2outputs(raw.view(b, N, 4), z, rays_d[sl], mask.view(b, N),
                                                                       kwargs["bgcolor"].to(dev))
                ret = {"rgb": rgb_map, "alpha": acc_map, "depth": depth_map, "weights": weights}
            else:
                ret = {"human_rgb": raw[:, :3].reshape(b, N, 3), "human_density": raw[:, 3].reshape(b, N),
                       "human_rgbsigma": raw.view(b, N, 4), "newsmpl_pts": pts, "pts_mask": mask.view(b, N)}

            def fwd_branch(c_pts, Rf_, Tf_, cond_, rows_dev=None):
                if grad:
                    d_ = ops.lbs_forward_ad(c_pts, vol_cl, Rf_, Tf_, bmin, bscale, K, rows_dev=rows_dev)
                    return _NonRigidFn.apply(self._token, self, "nrf", d_, cond_, band_w, rows_dev)
                d_ = ops.lbs_forward(c_pts, Rf_, Tf_, vol_cl, bmin, bscale, K, rows_dev=rows_dev)
                return self._nonrigid_fwd(self._nrf, d_, cond_, band_w, save=False, rows_dev=rows_dev)[0]

            if flow:                                                               # N:474-502
                ret["deform_pts_prev_final"] = fwd_branch(cnl_uses.pop(), R_fp, T_fp, cond_prev).view(b, N, 3)
            # N:505-536 (data-dependent size); the frame loops of eval.py never read the cycle outputs and switch them off
            if with_cycle and static_cycle:
                # fixed-capacity form for captured training steps: the selection is a device-side compaction, the row count
                # stays in device memory (`cycle_count`), rows past it are zero and receive zero gradients
                if B > chunk:
                    raise ValueError("static_cycle needs the whole ray batch in one chunk (cfg.chunk >= number of rays)")
                sel_cnl, observe, _, count = ops.compact_rows(mask, 0.005, cnl_uses.pop(), pts)
                # the kernels stop at `count` rows (the two-GEMM debug path of the backward, HOS_FUSED_BWD=0, has no row
                # limit and runs over the zero-padded capacity instead)
                ret["deform_pts_final"] = fwd_branch(sel_cnl, R_f, T_f, cond, rows_dev=count if ops.FUSED_THIN_BWD else None)
                ret["observe_pts"] = observe
                ret["cycle_count"] = count
            elif with_cycle:
                sel = torch.nonzero(mask.detach() > 0.005).reshape(-1)
                if sel.numel() > 0:
                    ret["deform_pts_final"] = fwd_branch(cnl_uses.pop().index_select(0, sel), R_f, T_f, cond)
                    ret["observe_pts"] = pts.view(-1, 3).index_select(0, sel)
                else:
                    ret["deform_pts_final"] = pts[0, 0][None]
                    ret["observe_pts"] = pts[0, 0][None]
            if not flow and self.stage != 2:
                ret["z_vals"] = z
                ret["rays_d"] = rays_d[sl]
            for k, v in ret.items():
                outs.setdefault(k, []).append(v)
        all_ret = {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in outs.items()}
        if self.stage != 2:
            all_ret["bgcolor"] = kwargs.get("bgcolor")                               # N:696
        return all_ret


class _NonRigidFn(torch.autograd.Function):
    """xyz = x + NonRigidMLP([cond | hann(x)]); parameter gradients go straight into the flat gradient buffer."""

    @staticmethod
    def forward(ctx, token, net: Network, which: str, x, cond, band_w, rows_dev=None):
        specs = net._nr if which == "nr" else net._nrf
        x = x.contiguous()
        xyz, saved = net._nonrigid_fwd(specs, x, cond, band_w, save=True, rows_dev=rows_dev)
        ctx.net, ctx.specs, ctx.saved, ctx.x, ctx.band_w, ctx.rows_dev = net, specs, saved, x, band_w, rows_dev
        ctx.mode = ops.get_gemm_mode()           # the backward pass (another host thread) runs in the arithmetic of this forward
        return xyz

    @staticmethod
    def backward(ctx, g):
        with ops.gemm_mode(ctx.mode):
            g_x = ctx.net._nonrigid_bwd(ctx.specs, ctx.saved, ctx.x, ctx.band_w, g.contiguous(), rows_dev=ctx.rows_dev)
        ctx.saved = None
        return None, None, None, g_x, None, None, None


class _CanonicalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, token, net: Network, cnl, state):
        cnl = cnl.contiguous()
        raw, saved = net._canonical_fwd(cnl, state, save=True)
        ctx.net, ctx.saved, ctx.cnl, ctx.raw, ctx.state = net, saved, cnl, raw, state
        ctx.mode = ops.get_gemm_mode()
        return raw

    @staticmethod
    def backward(ctx, g):
        with ops.gemm_mode(ctx.mode):
            g_cnl = ctx.net._canonical_bwd(ctx.saved, ctx.cnl, ctx.raw, g.contiguous(), ctx.state)
        ctx.saved = ctx.raw = None                       # break the output -> grad_fn -> ctx -> output cycle right away
        return None, None, g_cnl, None
