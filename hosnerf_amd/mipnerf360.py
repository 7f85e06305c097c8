"""State-conditional mip-NeRF-360 background renderer on MI355X.

Drop-in mirror of the reference classes (same names, constructor arguments, forward signature,
state_dict keys) -- 3rd_Complete_HOSNeRF/src/model/mipnerf360/model.py:114-540 (S3) and
1st_State-Conditional_Scene/src/model/mipnerf360/model.py:27-461 (S1):

    MipNeRF360MLP, NeRFMLP, PropMLP            M:114-375
    MipNeRF360(basedir, **gin kwargs).forward(batch, train_frac, randomized, is_train, near, far)
        -> (renderings, ray_history)           M:418-540 / M1:331-461

Everything per-sample runs in HIP kernels of libhosrender.so (resample, encode, fp32-MFMA linear
layers, compositing); this file only sequences launches and owns the flat parameter storage.
There is no CPU path: tensors must live on the HIP device.
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .flat import FlatModule, FlatStore, Region
from .geopoly import generate_basis

try:  # the reference classes are @gin.configurable; keep that surface when gin is installed
    import gin  # type: ignore

    _configurable = gin.configurable
except Exception:  # gin is absent in the build image
    def _configurable(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

POS_FEATS = 504          # 2 * 12 levels * 21 directions
EMBED = 64
X_LD = 576               # 568 padded to a multiple of 32
VIEW_FEATS = 27
XV_LD = 288              # 256 bottleneck + 27 view features, padded to a multiple of 32


def select_state(time: float, transitions_times) -> int:
    """M:224-293: index of the state embedding for `time` (host-side, like the reference's python ifs)."""
    if transitions_times is None or len(transitions_times) == 0:
        return 0
    eps = 1e-5
    if time < transitions_times[0] - eps:
        return 0
    n = len(transitions_times)
    if n > 6:
        raise NotImplementedError("the reference supports at most 7 states (M:279-293)")
    for k in range(1, n):
        if time <= transitions_times[k] + eps:
            return k
    return n


class _Lin(nn.Module):
    """Holder that gives a (weight, bias) pair the reference's `nn.Linear` attribute names."""

    def __init__(self, weight: nn.Parameter, bias: nn.Parameter):
        super().__init__()
        self.weight = weight
        self.bias = bias


class _PlanesSaved:
    """Activations of one planes-mode MLP query kept for the backward pass."""

    __slots__ = ("X16", "Xb", "y16", "yb", "WT")

    def __init__(self, X16, Xb, y16, yb, WT):
        self.X16, self.Xb, self.y16, self.yb, self.WT = X16, Xb, y16, yb, WT


class _LayerSpec:
    __slots__ = ("W", "b", "N", "K", "Kpad", "Npad")

    def __init__(self, store: FlatStore, N: int, K: int):
        self.N, self.K = N, K
        self.Kpad, self.Npad = ops.round_up(K, 32), ops.round_up(N, 32)
        self.W = store.alloc(N, K, self.Npad, self.Kpad)
        self.b = store.alloc(1, N, 1, self.Npad)


@_configurable()
class MipNeRF360MLP(FlatModule):
    """M:114-351.  Parameter names: pts_linear.{i}.{weight,bias}, density_layer, bottleneck_layer,
    views_linear.0, rgb_layer, bkgd_stateembeds.{k}; buffer pos_basis_t."""

    def __init__(self, basedir, netdepth: int = 8, netwidth: int = 256, bottleneck_width: int = 256,
                 netdepth_condition: int = 1, netwidth_condition: int = 128, min_deg_point: int = 0,
                 max_deg_point: int = 12, skip_layer: int = 4, skip_layer_dir: int = 4,
                 num_rgb_channels: int = 3, num_density_channels: int = 1, deg_view: int = 4,
                 bottleneck_noise: float = 0.0, density_bias: float = -1.0, density_noise: float = 0.0,
                 rgb_premultiplier: float = 1.0, rgb_bias: float = 0.0, rgb_padding: float = 0.001,
                 basis_shape: str = "icosahedron", basis_subdivision: int = 2, disable_rgb: bool = False):
        super().__init__()
        for name, value in list(locals().items()):
            if name not in ("self", "__class__"):
                setattr(self, name, value)
        unsupported = (min_deg_point != 0 or max_deg_point != 12 or deg_view != 4 or netdepth_condition != 1
                       or num_rgb_channels != 3 or num_density_channels != 1 or bottleneck_noise != 0.0
                       or density_noise != 0.0 or rgb_premultiplier != 1.0 or rgb_bias != 0.0
                       or basis_shape != "icosahedron" or basis_subdivision != 2 or bottleneck_width != 256
                       or netwidth % 32 != 0 or netwidth_condition % 32 != 0)
        if unsupported:
            raise NotImplementedError("hosnerf_amd implements the configurations the reference ships "
                                      "(defaults of M:116-140); got a non-default encoder/head option")
        self.register_buffer("pos_basis_t", generate_basis(basis_shape, basis_subdivision))

        # state embeddings (M:159-172)
        tt_path = os.path.join(basedir, "transitions_times.json") if basedir is not None else None
        if tt_path is not None and os.path.exists(tt_path):
            with open(tt_path, "r") as f:
                infos = json.load(f)
            self.transitions_times = np.stack([np.array(infos[k]["time"], dtype=np.float32) for k in infos], axis=0)
            n_states = self.transitions_times.shape[0] + 1
        else:
            self.transitions_times = None
            n_states = 1

        st = self.store
        pos_size = POS_FEATS + EMBED
        self._layers: List[_LayerSpec] = []
        self._skip_consumers = set()
        fan = pos_size
        for i in range(netdepth):
            self._layers.append(_LayerSpec(st, netwidth, fan))
            if i % skip_layer == 0 and i > 0:
                fan = netwidth + pos_size
                self._skip_consumers.add(i + 1)
            else:
                fan = netwidth
        if (netdepth - 1) % skip_layer == 0 and netdepth - 1 > 0:
            raise NotImplementedError("skip concat feeding the heads is not used by any shipped config")
        # skip consumers see [h | x] with x padded to X_LD
        for i in self._skip_consumers:
            if i < netdepth:
                L = self._layers[i]
                assert L.K == netwidth + pos_size
        # heads: NeRF -> combined [bottleneck(256) ; density(1)] matrix so one GEMM serves both
        if disable_rgb:
            self._head = _LayerSpec(st, 1, netwidth)
        else:
            self._head = _LayerSpec(st, bottleneck_width + 1, netwidth)
            self._views = _LayerSpec(st, netwidth_condition, bottleneck_width + VIEW_FEATS)
            self._rgb = _LayerSpec(st, 3, netwidth_condition)
        self._embeds = st.alloc(n_states, EMBED)
        st.materialize()

        full = (slice(None), slice(None))
        lins = []
        for L in self._layers:
            lins.append(_Lin(st.bind(L.W, (slice(0, L.N), slice(0, L.K)), (L.N, L.K)),
                             st.bind(L.b, (0, slice(0, L.N)), (L.N,))))
        self.pts_linear = nn.ModuleList(lins)
        H = self._head
        if disable_rgb:
            self.density_layer = _Lin(st.bind(H.W, (slice(0, 1), slice(0, netwidth)), (1, netwidth)),
                                      st.bind(H.b, (0, slice(0, 1)), (1,)))
        else:
            bw = bottleneck_width
            self.density_layer = _Lin(st.bind(H.W, (slice(bw, bw + 1), slice(0, netwidth)), (1, netwidth)),
                                      st.bind(H.b, (0, slice(bw, bw + 1)), (1,)))
            self.bottleneck_layer = _Lin(st.bind(H.W, (slice(0, bw), slice(0, netwidth)), (bw, netwidth)),
                                         st.bind(H.b, (0, slice(0, bw)), (bw,)))
            V, R = self._views, self._rgb
            self.views_linear = nn.ModuleList([_Lin(st.bind(V.W, (slice(0, V.N), slice(0, V.K)), (V.N, V.K)),
                                                    st.bind(V.b, (0, slice(0, V.N)), (V.N,)))])
            self.rgb_layer = _Lin(st.bind(R.W, (slice(0, 3), slice(0, R.K)), (3, R.K)),
                                  st.bind(R.b, (0, slice(0, 3)), (3,)))
        self.bkgd_stateembeds = nn.ParameterList(
            [st.bind(self._embeds, (k, slice(None)), (EMBED,)) for k in range(n_states)])
        self.reset_parameters()
        self._token = torch.zeros(1, requires_grad=True)

    # ------------------------------------------------------------------ init (M:174-209)
    @torch.no_grad()
    def reset_parameters(self):
        def kaiming(w):
            nn.init.kaiming_uniform_(w)

        def default_bias(b, fan_in):
            bound = 1.0 / math.sqrt(fan_in)
            b.uniform_(-bound, bound)

        for lin in self.pts_linear:
            kaiming(lin.weight)
            default_bias(lin.bias, lin.weight.shape[1])
        kaiming(self.density_layer.weight)
        default_bias(self.density_layer.bias, self.density_layer.weight.shape[1])
        if not self.disable_rgb:
            for lin in (self.bottleneck_layer, self.views_linear[0], self.rgb_layer):
                kaiming(lin.weight)
                default_bias(lin.bias, lin.weight.shape[1])
        for e in self.bkgd_stateembeds:
            e.normal_()

    def _after_flat_move(self):
        self._token = torch.zeros(1, device=self.store.param.device, requires_grad=True)

    # ------------------------------------------------------------------ kernels sequencing
    def _w(self, L: _LayerSpec, grad: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        flat = self.store.grad if grad else self.store.param
        return L.W.view(flat), L.b.view(flat).view(-1)

    def _forward_impl(self, X: torch.Tensor, viewdirs: Optional[torch.Tensor], B: int, S: int, save: bool):
        """X [P, X_LD] encoded samples -> density [P], rgb [P,3] | None, saved activations."""
        if isinstance(X, tuple) or self._use_planes():
            return self._forward_planes(X, viewdirs, B, S, save)
        P = X.shape[0]
        dev = X.device
        W = self.netwidth
        acts: List[torch.Tensor] = []
        h = X
        ping = [None, None]
        for i, L in enumerate(self._layers):
            if save:
                out = torch.empty(P, W, device=dev)
            else:
                if ping[i & 1] is None:
                    ping[i & 1] = torch.empty(P, W, device=dev)
                out = ping[i & 1]
            Wt, bt = self._w(L)
            if i in self._skip_consumers:
                ops.linear_fwd(h, W, Wt, bt, W, out, ops.EPI_RELU, A1=X, K1=X_LD)
            else:
                ops.linear_fwd(h, L.Kpad, Wt, bt, W, out, ops.EPI_RELU)
            if save:
                acts.append(out)
            h = out
        density = torch.empty(P, device=dev)
        Hs = self._head
        Wt, bt = self._w(Hs)
        if self.disable_rgb:
            ops.linear_fwd(h, W, Wt, bt, 1, None, ops.EPI_DENSITY, aux=density, p0=self.density_bias)
            return density, None, (X, acts)
        bw = self.bottleneck_width
        Xv = torch.empty(P, XV_LD, device=dev)
        ops.linear_fwd(h, W, Wt, bt, bw + 1, Xv, ops.EPI_NERF_HEAD, aux=density, aux_col=bw, p0=self.density_bias)
        ops.encode_viewdirs(viewdirs, S, Xv, bw)
        hv = torch.empty(P, self.netwidth_condition, device=dev)
        Wt, bt = self._w(self._views)
        ops.linear_fwd(Xv, XV_LD, Wt, bt, self.netwidth_condition, hv, ops.EPI_RELU)
        rgb = torch.empty(P, 3, device=dev)
        Wt, bt = self._w(self._rgb)
        ops.linear_fwd(hv, self.netwidth_condition, Wt, bt, 3, rgb, ops.EPI_RGB, p0=self.rgb_padding)
        return density, rgb, (X, acts, Xv, hv)

    def _backward_impl(self, saved, density, rgb, g_density, g_rgb, state: int):
        """Accumulate parameter gradients into the flat grad buffer (fused wgrad accumulation)."""
        if isinstance(saved[0], _PlanesSaved):
            return self._backward_planes(saved, density, rgb, g_density, g_rgb, state)
        X, acts = saved[0], saved[1]
        P = X.shape[0]
        dev = X.device
        W = self.netwidth
        Hs = self._head
        g_density = None if g_density is None else g_density.contiguous().view(-1)
        if self.disable_rgb:
            dyh = torch.empty(P, 32, device=dev)
            ops.head_grad_padded(g_density, density, None, None, 0.0, dyh, 0, None)     # writes the zero padding too
        else:
            Xv, hv = saved[2], saved[3]
            bw = self.bottleneck_width
            NC = self.netwidth_condition
            g_rgb = None if g_rgb is None else g_rgb.contiguous().view(-1, 3)
            # operand rows [bottleneck gradient (written by the view layer's dgrad below) | density | 0 ...] and [rgb 3 | 0 ...]:
            # the padding columns are written by the head-gradient launch, not by a fill of the whole [P, 288] / [P, 32] matrices
            dz_rgb = torch.empty(P, 32, device=dev)
            dyh = torch.empty(P, XV_LD, device=dev)
            ops.head_grad_padded(g_density, density, g_rgb, rgb, self.rgb_padding, dyh, bw, dz_rgb)
            # rgb layer (M:344)
            Wt, _ = self._w(self._rgb)
            gW, gb = self._w(self._rgb, grad=True)
            ops.linear_wgrad(dz_rgb, hv, gW, gb, 3, NC)
            dzv = torch.empty(P, NC, device=dev)
            ops.linear_dgrad(dz_rgb, Wt, 32, NC, dzv, mask_src=hv)
            # view-conditioned layer (M:337-342); only the bottleneck columns need a data gradient
            Wt, _ = self._w(self._views)
            gW, gb = self._w(self._views, grad=True)
            ops.linear_wgrad(dzv, Xv, gW, gb, NC, XV_LD)
            ops.linear_dgrad(dzv, Wt, NC, bw, dyh)
        # head: [bottleneck ; density] (M:305, M:325)
        Wt, _ = self._w(Hs)
        gW, gb = self._w(Hs, grad=True)
        h_last = acts[-1]
        ops.linear_wgrad(dyh, h_last, gW, gb, Hs.N, W)
        dz = torch.empty(P, W, device=dev)
        ops.linear_dgrad(dyh, Wt, Hs.Npad, W, dz, mask_src=h_last)
        # trunk, last layer first
        embed_cols = []   # (temporary bias-grad, weight region, first embedding column)
        tmps = ops.zeros_many([(self._layers[0].Npad,)] * (1 + len(self._skip_consumers)), dev)      # one launch for the bias-gradient temporaries
        for i in range(len(self._layers) - 1, -1, -1):
            L = self._layers[i]
            Wt, _ = self._w(L)
            gW, gb = self._w(L, grad=True)
            inp = acts[i - 1] if i > 0 else X
            if i in self._skip_consumers:
                tmp = tmps.pop()
                ops.linear_wgrad(dz, inp, gW, tmp, W, W)
                ops.linear_wgrad(dz, X, gW, None, W, X_LD, w_col0=W)
                embed_cols.append((tmp, Wt, W + POS_FEATS, gb))
            elif i == 0:
                tmp = tmps.pop()
                ops.linear_wgrad(dz, X, gW, tmp, W, X_LD)
                embed_cols.append((tmp, Wt, POS_FEATS, gb))
            else:
                ops.linear_wgrad(dz, inp, gW, gb, W, W)
            if i > 0:
                dz_prev = torch.empty(P, W, device=dev)
                ops.linear_dgrad(dz, Wt, W, W, dz_prev, mask_src=inp)
                dz = dz_prev
        # state-embedding gradient: the 64 embedding columns of x are constant over samples, so
        # d embed = (sum_p dZ[p,:]) @ W[:, embed cols] = db @ W[:, embed cols]   (M:295-296)
        g_embed = self._embeds.view(self.store.grad)[state]
        for tmp, Wt, c0, gb in embed_cols:       # gb += db and g_embed += db @ W[:, embed columns], one launch per layer
            ops.state_embed_grad(tmp, Wt, c0, W, gb, g_embed)

    # ------------------------------------------------------------------ planes path (ops.GEMM_PLANES)
    PLANES_MIN_WIDTH = int(os.environ.get("HOS_PLANES_MIN_WIDTH", "256"))

    def _use_planes(self) -> bool:
        return ops.get_gemm_mode() == ops.GEMM_PLANES and self.netwidth >= self.PLANES_MIN_WIDTH

    # Round 5: ONE activation format (bf16 planes) for an MLP whose outputs are only rendered.  The proposal MLPs keep fp16
    # (hi, lo) forward planes: their densities steer the resampling, and 22-bit operands are what keeps `bin_idx` on the
    # reference's values; the NeRF MLP's density and colour go straight into the composite, where bf16 pairs
    # (2^-17 per product) cost ~3e-5 RGB L-inf.  Its layers then write their output once -- forward operand, ReLU
    # mask and weight-gradient operand are the same planes -- and there is no fp16 range to guard.  HOS_NERF_BF16_FWD=0: fp16.
    BF16_FWD = False

    def _bf16_fwd(self) -> bool:
        return self.BF16_FWD and os.environ.get("HOS_NERF_BF16_FWD", "1") != "0"

    def _weight_planes(self, need_t: bool):
        """fp16 hi/lo planes of every weight (one pass over this MLP's span of the flat buffer: the planes keep the
        flat layout, so a layer's planes are a view at its region offset) and, for the backward pass, transposed
        bf16 planes [Kpad][Npad] of the trunk / head weights (dgrad operand)."""
        st = self.store
        specs = list(self._layers) + [self._head]
        lo = min(L.W.offset for L in specs)
        hi = max(L.W.offset + L.W.numel for L in specs)
        span = st.param[lo:hi].view(1, -1)
        if self._bf16_fwd():
            _, w16 = ops.split_planes2(span, ld=hi - lo, want16=False)
        else:
            w16, _ = ops.split_planes2(span, ld=hi - lo, wantb=False)
        flat16 = w16.t.view(-1)

        def view16(L):
            return ops.Planes.from_flat(flat16, L.W.offset - lo, L.Npad, L.Kpad)

        W16 = [view16(L) for L in specs]
        WT = None
        if need_t:
            WT = ops.split_planes_T_batch([L.W.view(st.param) for L in specs])        # one launch for the whole MLP
        return W16, WT

    def _forward_planes(self, X, viewdirs, B: int, S: int, save: bool):
        """X: (fp16 Planes, bf16 Planes | None) from hos_encode_ipe_planes, or an fp32 [P, X_LD] tensor (split here)."""
        W = self.netwidth
        W16, WT = self._weight_planes(need_t=save)
        one_fmt = self._bf16_fwd()
        if isinstance(X, tuple):
            X16, Xb = X
            if (save or one_fmt) and Xb is None:
                raise ValueError("the backward pass / the bf16-only forward needs the bf16 planes of the encoding")
        else:
            X16, Xb = ops.split_planes2(X, C=X_LD, ld=X_LD, want16=not one_fmt, wantb=save or one_fmt)
        if one_fmt:
            X16 = Xb            # the forward operand IS the bf16 planes
        P = X16.rows
        dev = X16.t.device
        h = X16
        kin = X_LD
        y16: List[ops.Planes] = []
        yb: List[ops.Planes] = []
        ping = [None, None]
        fdt = torch.bfloat16 if one_fmt else torch.float16
        for i, L in enumerate(self._layers):
            if save:
                out = ops.Planes.empty(P, W, fdt, dev, relu_bits=True)      # + 1 bit per element: the ReLU mask of the backward pass
                outb = out if one_fmt else ops.Planes.empty(P, W, torch.bfloat16, dev)
            else:
                if ping[i & 1] is None:
                    ping[i & 1] = ops.Planes.empty(P, W, fdt, dev)
                out, outb = ping[i & 1], (ping[i & 1] if one_fmt else None)
            _, bt = self._w(L)
            Yf, Ybf = (None, out) if one_fmt else (out, outb)
            if i in self._skip_consumers:
                ops.linearp_fwd(h, W, W16[i], bt, P, W, True, Yf, Ybf, A1=X16, K1=X_LD)
            else:
                ops.linearp_fwd(h, kin, W16[i], bt, P, W, True, Yf, Ybf)
            if save:
                y16.append(out)
                yb.append(outb)
            h, kin = out, W
        density = torch.empty(P, device=dev)
        Hs = self._head
        _, bt = self._w(Hs)
        saved0 = _PlanesSaved(X16, Xb, y16, yb, WT)
        Wh, _ = self._w(Hs)
        if self.disable_rgb:
            # Linear(width, 1) + softplus: one pass over the activation planes (a GEMM tile for one column cost 127 us per
            # 262 144 rows against 268 MB of input)
            if ops.ROWDOT_HEADS:
                ops.planes_rowdot(h, W, Wh[0], bt[0:1], density, p0=self.density_bias)
            else:
                ops.linearp_fwd(h, W, W16[-1], bt, P, 1, False, None, None, epilogue=ops.EPI_DENSITY, aux=density,
                                p0=self.density_bias)
            return density, None, (saved0,)
        bw = self.bottleneck_width
        Xv = torch.empty(P, XV_LD, device=dev)
        if ops.ROWDOT_HEADS:         # [bottleneck 256 | density 1]: the 256 columns as ONE column tile, the density column as a row dot
            ops.linearp_fwd(h, W, W16[-1], bt, P, bw, False, None, None, C=Xv, epilogue=ops.EPI_NONE)
            ops.planes_rowdot(h, W, Wh[bw], bt[bw:bw + 1], density, p0=self.density_bias)
        else:
            ops.linearp_fwd(h, W, W16[-1], bt, P, bw + 1, False, None, None, C=Xv, epilogue=ops.EPI_NERF_HEAD,
                            aux=density, aux_col=bw, p0=self.density_bias)
        ops.encode_viewdirs(viewdirs, S, Xv, bw)
        hv = torch.empty(P, self.netwidth_condition, device=dev)
        Wt, bt = self._w(self._views)
        ops.linear_fwd(Xv, XV_LD, Wt, bt, self.netwidth_condition, hv, ops.EPI_RELU)
        rgb = torch.empty(P, 3, device=dev)
        Wt, bt = self._w(self._rgb)
        ops.linear_fwd(hv, self.netwidth_condition, Wt, bt, 3, rgb, ops.EPI_RGB, p0=self.rgb_padding)
        return density, rgb, (saved0, None, Xv, hv)

    def _backward_planes(self, saved, density, rgb, g_density, g_rgb, state: int):
        sv: _PlanesSaved = saved[0]
        P = sv.X16.rows
        dev = density.device
        W = self.netwidth
        Hs = self._head
        nl = len(self._layers)
        g_density = None if g_density is None else g_density.contiguous().view(-1)
        if self.disable_rgb:
            dyh = torch.empty(P, 32, device=dev)
            ops.head_grad_padded(g_density, density, None, None, 0.0, dyh, 0, None)     # writes the zero padding too
        else:
            Xv, hv = saved[2], saved[3]
            bw = self.bottleneck_width
            NC = self.netwidth_condition
            g_rgb = None if g_rgb is None else g_rgb.contiguous().view(-1, 3)
            # operand rows [bottleneck gradient (written by the view layer's dgrad below) | density | 0 ...] and [rgb 3 | 0 ...]:
            # the padding columns are written by the head-gradient launch, not by a fill of the whole [P, 288] / [P, 32] matrices
            dz_rgb = torch.empty(P, 32, device=dev)
            dyh = torch.empty(P, XV_LD, device=dev)
            ops.head_grad_padded(g_density, density, g_rgb, rgb, self.rgb_padding, dyh, bw, dz_rgb)
            Wt, _ = self._w(self._rgb)
            gW, gb = self._w(self._rgb, grad=True)
            ops.linear_wgrad(dz_rgb, hv, gW, gb, 3, NC)
            dzv = torch.empty(P, NC, device=dev)
            ops.linear_dgrad(dz_rgb, Wt, 32, NC, dzv, mask_src=hv)
            Wt, _ = self._w(self._views)
            gW, gb = self._w(self._views, grad=True)
            ops.linear_wgrad(dzv, Xv, gW, gb, NC, XV_LD)
            ops.linear_dgrad(dzv, Wt, NC, bw, dyh)
        # head: [bottleneck ; density] (M:305, M:325) -- from here on everything is bf16 planes
        _, dyhP = ops.split_planes2(dyh, C=dyh.shape[1], ld=dyh.shape[1], want16=False)
        gW, gb = self._w(Hs, grad=True)
        ops.linearp_wgrad(dyhP, sv.yb[-1], gW, gb, P, Hs.N, W)
        dz = ops.Planes.empty(P, W, torch.bfloat16, dev)
        ops.linearp_dgrad(dyhP, sv.WT[-1], Hs.Npad, P, W, mask=sv.y16[-1], dX=dz)
        embed_cols = []
        tmps = ops.zeros_many([(self._layers[0].Npad,)] * (1 + len(self._skip_consumers)), dev)      # one launch for the bias-gradient temporaries
        for i in range(nl - 1, -1, -1):
            L = self._layers[i]
            Wt, _ = self._w(L)
            gW, gb = self._w(L, grad=True)
            inp_b = sv.yb[i - 1] if i > 0 else sv.Xb
            if i in self._skip_consumers:
                tmp = tmps.pop()
                ops.linearp_wgrad(dz, inp_b, gW, tmp, P, W, W)
                ops.linearp_wgrad(dz, sv.Xb, gW, None, P, W, X_LD, w_col0=W)
                embed_cols.append((tmp, Wt, W + POS_FEATS, gb))
            elif i == 0:
                tmp = tmps.pop()
                ops.linearp_wgrad(dz, sv.Xb, gW, tmp, P, W, X_LD)
                embed_cols.append((tmp, Wt, POS_FEATS, gb))
            else:
                ops.linearp_wgrad(dz, inp_b, gW, gb, P, W, W)
            if i > 0:
                dz_prev = ops.Planes.empty(P, W, torch.bfloat16, dev)
                ops.linearp_dgrad(dz, sv.WT[i], W, P, W, mask=sv.y16[i - 1], dX=dz_prev)
                dz = dz_prev
        g_embed = self._embeds.view(self.store.grad)[state]
        for tmp, Wt, c0, gb in embed_cols:       # gb += db and g_embed += db @ W[:, embed columns], one launch per layer
            ops.state_embed_grad(tmp, Wt, c0, W, gb, g_embed)

    # ------------------------------------------------------------------ reference-style forward
    def forward(self, gaussians, viewdirs, randomized, is_train, time):
        """M:311-351 signature.  `gaussians` must be the (tdist, rays_o, rays_d, radii) tuple produced by
        MipNeRF360.forward -- the cast/contract/IPE chain is fused with the encoder kernel, so the
        (means, covs) tensors of the reference are never materialised."""
        tdist, rays_o, rays_d, radii = gaussians
        return self.query(tdist, rays_o, rays_d, radii, viewdirs, float(time))

    def _zero_rgb(self, B: int, S: int, dev):
        """The all-zero colours a proposal MLP reports (M:347-349, `disable_rgb`): a constant, cached per shape (read-only)."""
        key = (B, S, str(dev))
        cache = getattr(self, "_zero_rgb_cache", None)
        if cache is None or cache[0] != key:
            self._zero_rgb_cache = cache = (key, torch.zeros(B, S, 3, device=dev))
        return cache[1]

    def query(self, tdist, rays_o, rays_d, radii, viewdirs, time: float) -> Dict[str, torch.Tensor]:
        B, S = tdist.shape[0], tdist.shape[1] - 1
        state = select_state(time, self.transitions_times)
        embed = self._embeds.view(self.store.param)[state]
        if self._use_planes():      # the encoder writes the 16-bit planes the trunk consumes (no fp32 copy of X)
            one_fmt = self._bf16_fwd()
            X = ops.encode_ipe_planes(tdist, rays_o, rays_d, radii, self.pos_basis_t, embed, X_LD,
                                      want_bf16=torch.is_grad_enabled() or one_fmt, want_fp16=not one_fmt)
        else:
            X = ops.encode_ipe(tdist, rays_o, rays_d, radii, self.pos_basis_t, embed, X_LD)
        if torch.is_grad_enabled():
            density, rgb = _MLPFn.apply(self._token, self, X, viewdirs, B, S, state)
        else:
            density, rgb, _ = self._forward_impl(X, viewdirs, B, S, save=False)
        density = density.view(B, S)
        rgb = self._zero_rgb(B, S, density.device) if self.disable_rgb else rgb.view(B, S, 3)
        return {"density": density, "rgb": rgb}


class _MLPFn(torch.autograd.Function):
    """Autograd node for one MLP query.  Parameter gradients are accumulated straight into the flat
    gradient buffer by the wgrad kernels (`p.grad` of every parameter is a view of it)."""

    @staticmethod
    def forward(ctx, token, mlp: MipNeRF360MLP, X, viewdirs, B, S, state):
        density, rgb, saved = mlp._forward_impl(X, viewdirs, B, S, save=True)
        ctx.mlp, ctx.saved, ctx.state = mlp, saved, state
        ctx.mode = ops.get_gemm_mode()           # the backward pass (another host thread) runs in the arithmetic of this forward
        ctx.density, ctx.rgb = density, rgb
        if rgb is None:
            rgb = torch.zeros(0, device=density.device)
            ctx.mark_non_differentiable(rgb)
        return density, rgb

    @staticmethod
    def backward(ctx, g_density, g_rgb):
        mlp = ctx.mlp
        with ops.gemm_mode(ctx.mode):
            mlp._backward_impl(ctx.saved, ctx.density, ctx.rgb, g_density, None if ctx.rgb is None else g_rgb, ctx.state)
        ctx.saved = ctx.density = ctx.rgb = None        # break the output -> grad_fn -> ctx -> output cycle right away
        return None, None, None, None, None, None, None


@_configurable()
class NeRFMLP(MipNeRF360MLP):
    """M:354-362."""
    BF16_FWD = True

    def __init__(self, basedir, netdepth: int = 8, netwidth: int = 1024):
        super().__init__(basedir, netdepth=netdepth, netwidth=netwidth)


@_configurable()
class PropMLP(MipNeRF360MLP):
    """M:365-375."""

    def __init__(self, basedir, netdepth: int = 4, netwidth: int = 256):
        super().__init__(basedir, netdepth=netdepth, netwidth=netwidth, disable_rgb=True)


@_configurable()
class MipNeRF360(FlatModule):
    """M:378-540 (S3) / M1:291-461 (S1) three-level proposal/NeRF driver.

    `render_levels=True` reproduces stage 1 (`renderings[l]["rgb"]`); `False` stage 3 (`renderings == []`).
    """

    def __init__(self, basedir, num_prop_samples: int = 64, num_nerf_samples: int = 32, num_levels: int = 3,
                 bg_intensity_range: Tuple[float, float] = (1.0, 1.0), anneal_slope: int = 10,
                 stop_level_grad: bool = True, use_viewdirs: bool = True, ray_shape: str = "cone",
                 disable_integration: bool = False, single_jitter: bool = True, dilation_multiplier: float = 0.5,
                 dilation_bias: float = 0.0025, num_glo_features: int = 0, num_glo_embeddings: int = 1000,
                 learned_exposure_scaling: bool = False, near_anneal_rate: Optional[float] = None,
                 near_anneal_init: float = 0.95, single_mlp: bool = False, resample_padding: float = 0.0,
                 use_gpu_resampling: bool = False, opaque_background: bool = False, render_levels: bool = True,
                 train_proposals: Optional[bool] = None):
        super().__init__()
        # Stage 3 (render_levels=False) has no interlevel loss and the resampling is detached (stop_level_grad): the proposal
        # MLPs receive no gradient there, so their queries need not keep anything for a backward pass.
        if train_proposals is None:
            train_proposals = render_levels
        for name, value in list(locals().items()):
            if name not in ("self", "__class__"):
                setattr(self, name, value)
        if (not stop_level_grad or not use_viewdirs or ray_shape != "cone" or disable_integration
                or not single_jitter or near_anneal_rate is not None or single_mlp
                or bg_intensity_range[0] != bg_intensity_range[1] or num_prop_samples > 64 or num_nerf_samples > 64):
            raise NotImplementedError("hosnerf_amd implements the configuration the reference ships (M:380-404 defaults)")
        mlps = [PropMLP(basedir) for _ in range(num_levels - 1)] + [NeRFMLP(basedir)]
        # merge the three stores into one flat buffer (one Adam launch / one all-reduce per step)
        host = {id(m): m.store.param.clone() for m in mlps}
        for m in mlps:
            base = self.store.adopt(m.store)
        self.store.materialize()
        for m in mlps:
            old = host[id(m)]
            first = m.store.regions[0].offset
            self.store.param[first:first + old.numel()].copy_(old)
            object.__setattr__(m, "store", self.store)
        self.store.rebind()
        self.mlps = nn.ModuleList(mlps)

    gemm_mode = None      # None: the process default (ops.set_gemm_mode); ops.GEMM_* pins the arithmetic of this module's GEMMs

    def lazy_param_spans(self):
        """[(offset, numel)] of the flat buffer that can be WITHOUT a gradient in a training step: every state embedding (one state
        per call, M:224-296 -- the other states' embeddings have `grad None` in the reference and its Adam skips them)."""
        return [(m._embeds.offset + k * EMBED, EMBED) for m in self.mlps for k in range(len(m.bkgd_stateembeds))]

    def _level0(self, B: int, dev):
        """The level-0 histogram of M:446-448 (constants): cached per (B, device) instead of three fills + a cat per call."""
        key = (B, str(dev))
        cache = getattr(self, "_level0_cache", None)
        if cache is None or cache[0] != key:
            sd = torch.cat([torch.zeros(B, 1, device=dev), torch.ones(B, 1, device=dev)], dim=-1)
            self._level0_cache = cache = (key, sd, torch.ones(B, 1, device=dev))
        return cache[1], cache[2]

    def forward(self, batch, train_frac, randomized, is_train, near, far, jitters=None, want_index: bool = False):
        """`self.gemm_mode` (None = the process default, ops.set_gemm_mode) selects the arithmetic of THIS module's GEMMs;
        without autograd the fp16 range guard may set it to exact fp32 (ops.guarded_forward)."""
        return ops.guarded_forward(self, batch["rays_o"].device,
                                   lambda: self._forward(batch, train_frac, randomized, is_train, near, far, jitters, want_index))

    def _forward(self, batch, train_frac, randomized, is_train, near, far, jitters=None, want_index: bool = False):
        rays_o = batch["rays_o"].contiguous()
        rays_d = batch["rays_d"].contiguous()
        viewdirs = batch["viewdirs"].contiguous()
        radii = batch["radii"].contiguous()
        B = rays_o.shape[0]
        dev = rays_o.device
        times = batch["times"]
        # the reference branches on `time` in python (M:230) = one host sync per call; pass a python float to avoid it
        time = float(times.reshape(-1)[0]) if isinstance(times, torch.Tensor) else float(times)

        sdist, weights = self._level0(B, dev)                                    # M:446-448: sdist = [0, 1], weights = [1]
        prod = 1
        # M:459-460.  `train_frac` may be a 1-element device tensor: the anneal factor is then evaluated inside the resampling
        # kernel, so a step captured once in a hipGraph keeps annealing as training advances (no host value baked in)
        frac_dev = train_frac if isinstance(train_frac, torch.Tensor) and train_frac.is_cuda else None
        if frac_dev is not None:
            anneal = 1.0
        else:
            train_frac = float(train_frac)
            anneal = (self.anneal_slope * train_frac) / ((self.anneal_slope - 1) * train_frac + 1) if self.anneal_slope > 0 else 1.0
        ray_history, renderings = [], []
        for lvl in range(self.num_levels):
            is_prop = lvl < self.num_levels - 1
            S = self.num_prop_samples if is_prop else self.num_nerf_samples
            dilation = self.dilation_bias + self.dilation_multiplier * 1.0 / prod      # M:450-455
            prod *= S
            jit = None if jitters is None else jitters[lvl]
            out = ops.resample(sdist, weights, S, dilation, anneal, bool(randomized), near, far, jitter=jit,
                               resample_padding=self.resample_padding, want_index=want_index,
                               train_frac_dev=frac_dev, anneal_slope=float(self.anneal_slope))
            sdist, tdist = out[0], out[1]
            if is_prop and not self.train_proposals:
                with torch.no_grad():
                    res = self.mlps[lvl].query(tdist, rays_o, rays_d, radii, viewdirs, time)
            else:
                res = self.mlps[lvl].query(tdist, rays_o, rays_d, radii, viewdirs, time)
            weights = ops.alpha_weights(res["density"], tdist, rays_d, self.opaque_background)
            res["sdist"], res["tdist"], res["weights"] = sdist, tdist, weights
            if want_index:
                res["bin_idx"] = out[2]
            ray_history.append(res)
            if self.render_levels:
                renderings.append({"rgb": ops.volumetric_rendering(res["rgb"], weights, self.bg_intensity_range[0])})
        return renderings, ray_history
