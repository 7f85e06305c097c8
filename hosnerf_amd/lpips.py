"""LPIPS term of the stage-2 / stage-3 training loss on the MI355X (csrc/hos_lpips.hip + the library's GEMM entry points).

Mirrors `LPIPS(net='vgg')` of the reference's vendored package (3rd_Complete_HOSNeRF/third_parties/lpips/lpips.py:22-122,
pretrained_networks.py:97-135) as the training step uses it (src/model/mipnerf360/model.py:582-584, 1664-1678: frozen, eval mode,
`lpips_func(2 rgb - 1, 2 target - 1)` on the unpacked [N, 3, P, P] patches, mean over patches, weight 1.0 in
configs/default.yaml:97-101).  The ImageNet VGG-16 filters are torchvision's download and do not exist offline: `load_vgg16_features`
takes them as a state dict (`features.N.weight` or `N.weight`), the learned 1x1 calibration weights come from the reference's own
`third_parties/lpips/weights/v0.1/vgg.pth` (`load_lin`).  Without weights the module raises -- there is no CPU or torch fallback.

Everything is channel-last, so the unpacked patches [N, P, P, 3] are the network input as they are; only the prediction needs a
gradient (the filters are frozen): the backward pass is thirteen input-gradient GEMMs + col2im, no weight gradient."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import call, ptr

VGG16_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]      # features[0:30] (P:105-114)
TAP_AFTER_CONV = (1, 3, 6, 9, 12)            # relu1_2, relu2_2, relu3_3, relu4_3, relu5_3 = behind the 2nd, 4th, 7th, 10th, 13th convolution
CHNS = (64, 128, 256, 512, 512)
SCALE = (0.458, 0.448, 0.450)


def patch_ray_index(patch_masks: torch.Tensor) -> torch.Tensor:
    """int32 [N*P*P]: the ray of every patch pixel in patch order (model.py:41-50 `_unpack_imgs`: rays are stored patch after patch,
    row-major inside a patch, only where the mask is set), -1 where a pixel has no ray.  Once per item, outside the step."""
    m = patch_masks.reshape(-1).to(torch.int64)
    return torch.where(m > 0, torch.cumsum(m, 0) - 1, torch.full_like(m, -1)).to(torch.int32)


class LPIPS(nn.Module):
    def __init__(self):
        super().__init__()
        self._W: List[torch.Tensor] = []       # [Cout, Kpad] per convolution, reduction index tap * Cin + cin
        self._b: List[torch.Tensor] = []
        self._lin: List[torch.Tensor] = []

    # ------------------------------------------------------------------ weights
    def load_vgg16_features(self, sd: Dict[str, torch.Tensor], device="cuda"):
        """`sd`: torchvision `vgg16().features` state (keys `N.weight` / `N.bias`, optionally prefixed `features.`), N in
        0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28."""
        sd = {k[len("features."):] if k.startswith("features.") else k: v for k, v in sd.items()}
        self._W, self._b, idx, cin = [], [], 0, 3
        for v in VGG16_CFG:
            if v == "M":
                idx += 1
                continue
            w = sd[f"{idx}.weight"].detach().float()
            if tuple(w.shape) != (v, cin, 3, 3):
                raise ValueError(f"features.{idx}.weight has shape {tuple(w.shape)}, expected {(v, cin, 3, 3)}")
            kpad = ops.round_up(9 * cin, 32)
            wg = torch.zeros(v, kpad)
            wg[:, :9 * cin] = w.permute(0, 2, 3, 1).reshape(v, 9 * cin)          # [out][ky][kx][cin] = tap-major, like hos_im2col3x3
            self._W.append(wg.to(device).contiguous())
            self._b.append(sd[f"{idx}.bias"].detach().float().to(device).contiguous())
            cin = v
            idx += 2
        return self

    def load_lin(self, obj, device="cuda"):
        """The calibration weights: the reference's `weights/v0.1/vgg.pth` state (`lin{k}.model.1.weight` [1, C, 1, 1]) or the five
        vectors concatenated (1 472 floats)."""
        if isinstance(obj, dict):
            parts = [obj[f"lin{k}.model.1.weight"].detach().float().reshape(-1) for k in range(5)]
        else:
            flat = torch.as_tensor(obj, dtype=torch.float32).reshape(-1)
            if flat.numel() != sum(CHNS):
                raise ValueError("expected 64 + 128 + 256 + 512 + 512 calibration weights")
            parts = list(torch.split(flat, list(CHNS)))
        self._lin = [p_.to(device).contiguous() for p_ in parts]
        return self

    @classmethod
    def from_files(cls, vgg16_path: str, lin_path: str, device="cuda") -> "LPIPS":
        """`vgg16_path`: torchvision's `vgg16` checkpoint (`vgg16-397923af.pth`: keys `features.N.*`, the classifier is ignored);
        `lin_path`: the reference's `third_parties/lpips/weights/v0.1/vgg.pth`."""
        return cls().load_vgg16_features(torch.load(vgg16_path, map_location="cpu"), device).load_lin(torch.load(lin_path, map_location="cpu"), device)

    def ready(self) -> bool:
        return len(self._W) == 13 and len(self._lin) == 5

    # ------------------------------------------------------------------ the loss term
    def loss(self, rgb: torch.Tensor, target_patches: torch.Tensor, ray_idx: torch.Tensor, bgcolor: torch.Tensor) -> torch.Tensor:
        """mean_i LPIPS(2 unpack(rgb)_i - 1, 2 target_i - 1)  (model.py:1673-1676).  rgb [B, 3] rendered ray colours, target_patches
        [N, P, P, 3] in [0, 1], ray_idx = patch_ray_index(patch_masks), bgcolor [3] in 0..255 (the batch's key)."""
        if not self.ready():
            raise RuntimeError("LPIPS: load_vgg16_features() and load_lin() first (the ImageNet VGG-16 weights are not part of this repository)")
        return _LPIPSFn.apply(rgb, self, target_patches, ray_idx, bgcolor)


def _lib_part(Np: int) -> int:
    return _lib.load().hos_lpips_part_floats(Np)


def _vgg_forward(mod: LPIPS, x0: torch.Tensor, NI: int, P: int):
    """x0 [NI * P * P, 3] (scaled images) -> (taps [5], per-convolution inputs, pool inputs, spatial sizes)."""
    dev = x0.device
    h, H, C = x0, P, 3
    taps, conv_in, conv_out, pools, geo = [], [], [], [], []
    ci = 0
    with ops.gemm_mode(ops.GEMM_FP32):
        for v in VGG16_CFG:
            if v == "M":
                out = torch.empty(NI * (H // 2) * (H // 2), C, device=dev)
                call("hos_maxpool2x2_fwd", ptr(h), NI, H, H, C, ptr(out))
                pools.append((h, H, C))
                h, H = out, H // 2
                continue
            W, b = mod._W[ci], mod._b[ci]
            kpad = W.shape[1]
            col = torch.empty(NI * H * H, kpad, device=dev)
            call("hos_im2col3x3", ptr(h), NI, H, H, C, ptr(col), kpad)
            y = torch.empty(NI * H * H, v, device=dev)
            M = NI * H * H
            if ((M + 127) // 128) * ((v + 127) // 128) < 64 and kpad >= 1152:
                # few output tiles, long reduction (the 8 x 8 .. 2 x 2 feature maps): one tile kernel would walk 36-144 K tiles on
                # 4-16 workgroups (205 us per layer); the reduction is split over ~256 workgroups whose partial tiles are summed in a
                # FIXED order (slabs; round 4 used atomics and the gradient changed from run to run) together with bias + ReLU
                need = int(_lib.load().hos_linear_fwd_splitk_ws_floats(M, v, kpad))
                ws = ops._bwd_workspace(dev, need=max(need, 4))
                call("hos_linear_fwd_splitk_det", ptr(col), kpad, ptr(W), kpad, ptr(b), 1, ptr(y), v, M, v, kpad, ptr(ws), ws.numel())
            else:
                ops.linear_fwd(col, kpad, W, b, v, y, ops.EPI_RELU)
            conv_in.append(h)
            conv_out.append(y)
            geo.append((H, C, v))
            if ci in TAP_AFTER_CONV:
                taps.append((y, H * H, v))
            h, C = y, v
            ci += 1
    return taps, conv_in, conv_out, pools, geo


class _LPIPSFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, mod: LPIPS, target, ray_idx, bgcolor):
        Np, P = int(target.shape[0]), int(target.shape[1])
        npix = Np * P * P
        dev = rgb.device
        rgb = rgb.contiguous()
        pred = torch.empty(npix, 3, device=dev)
        call("hos_unpack_patches_fwd", ptr(rgb), ptr(ray_idx, torch.int32), ptr(bgcolor.contiguous()), 1.0 / 255.0, npix, ptr(pred))
        x0 = torch.empty(2 * npix, 3, device=dev)
        call("hos_lpips_prep", ptr(pred), npix, ptr(x0))
        call("hos_lpips_prep", ptr(target.contiguous()), npix, ptr(x0) + 4 * 3 * npix)
        taps, conv_in, conv_out, pools, geo = _vgg_forward(mod, x0, 2 * Np, P)
        part = ops.zeros(int(_lib_part(Np)), dev)
        for k, (f, HW, C) in enumerate(taps):
            call("hos_lpips_head_fwd", ptr(f), ptr(mod._lin[k]), Np, HW, C, 1.0 / (HW * Np), ptr(part))
        out = torch.empty(1, device=dev)
        call("hos_lpips_finish", ptr(part), Np, ptr(out))
        ctx.mod, ctx.saved = mod, (taps, conv_in, conv_out, pools, geo, ray_idx, Np, P, rgb.shape[0])
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        mod = ctx.mod
        taps, conv_in, conv_out, pools, geo, ray_idx, Np, P, B = ctx.saved
        dev = gout.device
        gscale = gout.reshape(1).contiguous().float()
        g = None                              # gradient w.r.t. the current activation, prediction images only
        ci, pi, ti = 12, len(pools) - 1, 4
        with ops.gemm_mode(ops.GEMM_FP32):
            for v in reversed(VGG16_CFG):
                if v == "M":
                    src, H, C = pools[pi]
                    pi -= 1
                    gi = torch.empty(Np * H * H, C, device=dev)
                    call("hos_maxpool2x2_bwd", ptr(g), ptr(src), Np, H, H, C, ptr(gi))
                    g = gi
                    continue
                H, Cin, Cout = geo[ci]
                if ci in TAP_AFTER_CONV:
                    f, HW, C = taps[ti]
                    if g is None:
                        g = torch.empty(Np * HW, C, device=dev)
                        acc = 0
                    else:
                        acc = 1
                    call("hos_lpips_head_bwd", ptr(f), ptr(mod._lin[ti]), Np, HW, C, 1.0 / (HW * Np), ptr(gscale), acc, ptr(g))
                    ti -= 1
                W = mod._W[ci]
                kpad = W.shape[1]
                dcol = torch.empty(Np * H * H, kpad, device=dev)
                ops.linear_dgrad(g, W, Cout, kpad, dcol)
                x_in = conv_in[ci]
                # the convolution's input is a ReLU output exactly when the layer below is a convolution (not the image, not a pool)
                relu_below = ci > 0 and VGG16_CFG[_cfg_pos(ci) - 1] != "M"
                gi = torch.empty(Np * H * H, Cin, device=dev)
                call("hos_col2im3x3", ptr(dcol), kpad, Np, H, H, Cin, ptr(x_in) if relu_below else None, ptr(gi))
                g = gi
                ci -= 1
        g_rgb = ops.zeros((B, 3), dev)
        call("hos_unpack_patches_bwd", ptr(g), ptr(ray_idx, torch.int32), Np * P * P, 2.0 / SCALE[0], 2.0 / SCALE[1], 2.0 / SCALE[2], ptr(g_rgb))
        ctx.saved = None
        return g_rgb, None, None, None, None


def _cfg_pos(ci: int) -> int:
    """Position of convolution `ci` in VGG16_CFG."""
    n = -1
    for pos, v in enumerate(VGG16_CFG):
        if v != "M":
            n += 1
            if n == ci:
                return pos
    raise IndexError(ci)
