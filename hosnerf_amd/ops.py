"""Python-side wrappers of the C-ABI kernels: shape checks, output allocation, autograd glue.

Nothing here computes on the CPU; every function enqueues HIP kernels of libhosrender.so on the
current torch stream.  Tensors are fp32, contiguous, on the HIP device.
"""
from __future__ import annotations

import os
import threading

import math
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr

EPS = 1.1920929e-07

EPI_NONE, EPI_RELU, EPI_DENSITY, EPI_RGB, EPI_NERF_HEAD, EPI_SIGMOID_RELU4, EPI_RESIDUAL = 0, 1, 2, 3, 4, 5, 6


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# ------------------------------------------------------------------------------------------ kernel timing
class KernelEvents:
    """Optional HIP-event timing of every GEMM launch (bench.py's live roofline measurement).

    Events are recorded on the current torch stream -- the stream the kernels are launched on -- so the
    elapsed time of a (start, stop) pair is that launch's device duration (plus launch gaps)."""

    def __init__(self):
        self.records = {}

    def record(self, key, flops, launch):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        launch()
        b.record()
        self.records.setdefault(key, []).append((a, b, flops))

    def summary(self):
        torch.cuda.synchronize()
        rows = []
        for key, recs in self.records.items():
            ms = [a.elapsed_time(b) for a, b, _ in recs]
            flops = recs[0][2]
            tot = sum(ms)
            rows.append({"kernel": key, "launches": len(recs), "total_ms": tot, "avg_us": 1e3 * tot / len(recs),
                         "flop_per_launch": flops, "tflops": flops * len(recs) / (tot * 1e-3) / 1e12 if tot > 0 else 0.0})
        rows.sort(key=lambda r: -r["total_ms"])
        return rows


_EVENTS: Optional[KernelEvents] = None


def set_kernel_events(ev: Optional[KernelEvents]):
    global _EVENTS
    _EVENTS = ev


def _timed(key, flops, launch):
    if _EVENTS is None:
        launch()
    else:
        _EVENTS.record(key, flops, launch)


# ------------------------------------------------------------------------------------------ GEMMs
GEMM_FP32, GEMM_BF16X3, GEMM_PLANES = 0, 1, 2
_default_mode = GEMM_PLANES  # process default: planes trunks + split GEMMs elsewhere (the library's own default is GEMM_BF16X3)
_tls = threading.local()     # .mode: this thread's override (None / absent = follow the default), see `gemm_mode`


def set_gemm_mode(mode: int):
    """The process DEFAULT (set once at start-up): 0 = exact fp32 MFMA; 1 = split precision (fp16 / bf16 hi-lo, 3 MFMAs per
    product) on fp32 operands; 2 = the same arithmetic, but the wide MLP trunks keep their activations as pre-split 16-bit
    planes (hos_linearp_*: LDS-DMA staging, no conversion work in the K loop); every other GEMM runs as in mode 1.
    Scoped switches go through the per-thread override (`gemm_mode`), not through this."""
    global _default_mode
    _default_mode = int(mode)
    _lib.check(_lib.load().hos_set_gemm_mode(GEMM_BF16X3 if _default_mode == GEMM_PLANES else _default_mode), "hos_set_gemm_mode")


def _set_thread_mode(mode: Optional[int]):
    _tls.mode = mode
    _lib.check(_lib.load().hos_set_thread_gemm_mode(-1 if mode is None else (GEMM_BF16X3 if mode == GEMM_PLANES else int(mode))),
               "hos_set_thread_gemm_mode")


def get_gemm_mode() -> int:
    """The mode the calling thread's next GEMM launches in (its override if one is active, the process default otherwise)."""
    m = getattr(_tls, "mode", None)
    return _default_mode if m is None else m


def linear_fwd(A0: torch.Tensor, K0: int, W: torch.Tensor, bias: Optional[torch.Tensor], N: int,
               out: Optional[torch.Tensor], epilogue: int = EPI_NONE, A1: Optional[torch.Tensor] = None,
               K1: int = 0, aux: Optional[torch.Tensor] = None, aux_col: int = -1, p0: float = 0.0,
               ldc: Optional[int] = None, M: Optional[int] = None, out_col0: int = 0, rows_dev: Optional[torch.Tensor] = None,
               relu_bits: Optional[torch.Tensor] = None):
    """out[M, out_col0:out_col0+N] = epi([A0[:, :K0] | A1[:, :K1]] @ W[:N, :K0+K1]^T + bias).  W is a [rows, ldw] buffer.
    For EPI_RESIDUAL `aux` is the residual matrix [M, >=N] (its row stride is passed as aux_col).
    `relu_bits` (`thin_relu_bits(M)`, only where `thin_dgrad_rows(M)` holds): receives the ReLU mask of the backward pass as one bit
    per output element, for `linear_dgrad(mask_bits=...)` of the layer above."""
    M = A0.shape[0] if M is None else M
    if epilogue == EPI_RESIDUAL:
        aux_col = aux.stride(0)
    if (THIN_GEMM and rows_dev is None and A1 is None and 128 < N <= 256 and K0 <= 320 and K0 % 4 == 0 and M >= 16384 and epilogue in (EPI_NONE, EPI_RELU)
            and ldc is None and out is not None and _lib.load().hos_get_gemm_mode() == GEMM_BF16X3):
        # many rows through a thin layer: persistent kernel with the weight in registers (hos_thin.hip)
        _timed(f"thin_fwd[M={M},N={N},K={K0}]", 2.0 * M * N * K0, lambda: call(
            "hos_thin_linear_fwd", ptr(A0), A0.stride(0), ptr(W), W.stride(0), ptr(bias), ptr(out) + 4 * out_col0, out.stride(0),
            M, N, K0, epilogue, ptr(relu_bits, torch.int16)))
        return out
    if relu_bits is not None:
        raise _lib.HosLibraryError("relu_bits: this layer does not run on the thin kernel (check ops.thin_dgrad_rows first)")
    _timed(f"gemm_fwd[M={M},N={N},K={K0 + K1}]", 2.0 * M * N * (K0 + K1), lambda: call(
        "hos_linear_fwd", ptr(A0), A0.stride(0), K0, ptr(A1), 0 if A1 is None else A1.stride(0), K1,
        ptr(W), W.stride(0), ptr(bias), ptr(out) + 4 * out_col0, (0 if out is None else out.stride(0)) if ldc is None else ldc,
        M, N, epilogue, ptr(aux), aux_col, float(p0), 0.0, ptr(rows_dev, torch.int32)))
    return out


# the canonical MLP with its per-call state embedding folded into the biases of the input layer and the skip layer (hos_thin.hip)
CNL_FOLD = os.environ.get("HOS_CNL_FOLD", "1") != "0"
_CNL_FOLD_WS = {}
_ZERO1 = {}


def zero1(device) -> torch.Tensor:
    """A cached [1] zero on the device (e.g. the pad column of an embedder row passed as a one-element 'state')."""
    key = str(torch.device(device))
    if key not in _ZERO1:
        _ZERO1[key] = torch.zeros(1, device=device)
    return _ZERO1[key]


def _table(ptrs, ctype=None):
    import ctypes
    ctype = ctypes.c_void_p if ctype is None else ctype
    return (ctype * len(ptrs))(*ptrs)


def clear_last_error() -> int:
    """Return and clear the runtime's sticky last-error code (hos_clear_last_error): call once after recovering from a failed hipGraph
    capture, before launching eagerly -- otherwise the next launch of this library reports the capture's error as its own."""
    return int(_lib.load().hos_clear_last_error())


def copy_or_zero_n(dsts, srcs=None):
    """dsts[i][...] = srcs[i] (None: zeros) for up to 8 contiguous fp32 tensors per launch (hos_copy_or_zero_n): fills of
    accumulation targets, stacks of small per-frame tensors -- one launch per group instead of one torch launch per tensor."""
    import ctypes
    for i0 in range(0, len(dsts), 8):
        d = dsts[i0:i0 + 8]
        sr = [None] * len(d) if srcs is None else srcs[i0:i0 + 8]
        for a, b in zip(d, sr):
            if b is not None and b.numel() != a.numel():
                raise _lib.HosLibraryError("copy_or_zero_n: size mismatch")
        call("hos_copy_or_zero_n", len(d), _table([ptr(t) for t in d]), _table([ptr(t) for t in sr]),
             _table([t.numel() for t in d], ctypes.c_longlong))


def zeros_many(shapes, device, dtype=torch.float32):
    """Zeroed fp32 tensors of the given shapes out of ONE allocation and ONE library launch (sizes rounded to 16 bytes so every
    view stays aligned)."""
    sizes = [(int(np.prod(sh)) + 3) // 4 * 4 for sh in shapes]
    buf = torch.empty(sum(sizes), device=device, dtype=dtype)
    copy_or_zero_n([buf])
    out, o = [], 0
    for sh, n in zip(shapes, sizes):
        out.append(buf[o:o + int(np.prod(sh))].view(sh))
        o += n
    return out


def zeros(shape, device):
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    return zeros_many([shape], device)[0]


def zero_(t: torch.Tensor):
    copy_or_zero_n([t])
    return t


def add_n(ts):
    """sum of up to 8 same-shaped contiguous fp32 tensors in one launch."""
    import ctypes
    out = torch.empty_like(ts[0])
    call("hos_add_n", len(ts), _table([ptr(t.contiguous()) for t in ts]), ts[0].numel(), ptr(out))
    return out


class _Fanout(torch.autograd.Function):
    """x -> n aliases of x for n consumers; the backward sums their cotangents in ONE launch (autograd's own accumulation is one
    add launch per extra consumer)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g.contiguous() for g in gs if g is not None]
        if not gs:
            return None, None
        return (gs[0] if len(gs) == 1 else add_n(gs)), None


def fanout(x: torch.Tensor, n: int):
    return _Fanout.apply(x, n) if (n > 1 and x.requires_grad and torch.is_grad_enabled()) else (x,) * n


def canonical_fold_pack(W0, b0, W5, b5, embed, n_out, nf, nh, out):
    """`out` = (W0f [n_out, nfp], b0f [n_out], W5f [n_out, nfp + nh], b5f [n_out]), see hos_canonical_fold_pack."""
    call("hos_canonical_fold_pack", ptr(W0), W0.stride(0), ptr(b0), ptr(W5), W5.stride(0), ptr(b5), ptr(embed), n_out, nf, embed.numel(), nh,
         ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]))


def canonical_fold_unfold(grads, W0, W5, embed, n_out, nf, nh, gW0, gb0, gW5, gb5, g_embed):
    """`grads` = (gW0f, db0, gW5f, db5) accumulated by the backward launches of the folded layers -> += into the reference-shaped
    gradients and the state embedding's (hos_canonical_fold_unfold)."""
    call("hos_canonical_fold_unfold", ptr(grads[0]), ptr(grads[1]), ptr(grads[2]), ptr(grads[3]), ptr(W0), W0.stride(0), ptr(W5), W5.stride(0),
         ptr(embed), n_out, nf, embed.numel(), nh, ptr(gW0), ptr(gb0), ptr(gW5), ptr(gb5), ptr(g_embed))


def cnl_fold_views(buf: torch.Tensor, n_out: int, nfp: int, nh: int):
    """(W [n_out, nfp], b [n_out], W5 [n_out, nfp + nh], b5 [n_out]) as views of one flat buffer (16-byte aligned matrices)."""
    o1 = n_out * nfp
    o2 = o1 + n_out
    o3 = o2 + n_out * (nfp + nh)
    return buf[:o1].view(n_out, nfp), buf[o1:o2], buf[o2:o3].view(n_out, nfp + nh), buf[o3:o3 + n_out]


def cnl_fold_grad_workspace(device, n_out: int, nfp: int, nh: int) -> torch.Tensor:
    """Gradient buffers of the two folded layers, one per (device, stream): filled by the backward launches, consumed by
    canonical_fold_unfold on the same stream."""
    key = _stream_key(device) + (n_out, nfp, nh)
    if key not in _CNL_FOLD_WS:
        _CNL_FOLD_WS[key] = torch.empty(n_out * (2 * nfp + nh + 2), device=device)
    return _CNL_FOLD_WS[key]


def thin_dgrad_rows(M: int) -> bool:
    """True when `linear_dgrad` of a <= 256-wide layer over M rows runs on the thin kernel (which takes W / mask windows at any
    column; the tiled GEMMs need 16-byte aligned windows)."""
    return bool(THIN_GEMM and M >= 16384 and _lib.load().hos_get_gemm_mode() == GEMM_BF16X3)


RELU_BITS = os.environ.get("HOS_RELU_BITS", "1") == "1"      # thin layers: ReLU mask of the backward pass as bits (A/B switch)


def thin_relu_bits(M: int, device) -> torch.Tensor:
    """Storage of a thin layer's ReLU bit mask: 16 bits per lane, 512 lanes per 32-row tile (opaque; hos_thin.hip)."""
    return torch.empty(((M + 31) // 32) * 512, dtype=torch.int16, device=device)


def linear_dgrad(dY: torch.Tensor, W: torch.Tensor, Npad: int, K: int, out: torch.Tensor,
                 mask_src: Optional[torch.Tensor] = None, accumulate: bool = False, w_col0: int = 0, mask_col0: int = 0,
                 thin: bool = False, mask_bits: Optional[torch.Tensor] = None):
    """out[M, :K] = (dY[:, :Npad] @ W[:Npad, w_col0:w_col0+K]) * (mask_src[:, mask_col0:mask_col0+K] > 0).
    `thin`: take the thin kernel also for K <= 128 output columns (by default those go to the tiled GEMM).
    `mask_bits`: the bit mask `linear_fwd(relu_bits=...)` wrote for the K <= 256 columns of this layer's input (thin kernel only);
    used instead of `mask_src`."""
    M = dY.shape[0]
    wptr = ptr(W) + 4 * w_col0
    mptr = None if mask_src is None else ptr(mask_src) + 4 * mask_col0
    if mask_bits is not None and K > 256:
        raise _lib.HosLibraryError("mask_bits cover one thin launch (K <= 256 columns)")
    if (not accumulate and Npad <= 256 and Npad % 4 == 0 and (K > 128 or thin) and thin_dgrad_rows(M)):
        # output columns in chunks of <= 256 (the skip layer's [P,384] input gradient is two launches)
        for k0 in range(0, K, 256):
            kc = min(256, K - k0)
            _timed(f"thin_dgrad[M={M},N={kc},K={Npad}]", 2.0 * M * kc * Npad, lambda: call(
                "hos_thin_linear_dgrad", ptr(dY), dY.stride(0), wptr + 4 * k0, W.stride(0), Npad,
                None if mptr is None else mptr + 4 * k0, 0 if mask_src is None else mask_src.stride(0), ptr(mask_bits, torch.int16),
                ptr(out) + 4 * k0, out.stride(0), M, kc))
        return out
    if mask_bits is not None and mask_src is None:
        raise _lib.HosLibraryError("mask_bits: this layer does not run on the thin kernel (check ops.thin_dgrad_rows first)")
    _timed(f"gemm_dgrad[M={M},N={K},K={Npad}]", 2.0 * M * K * Npad, lambda: call(
        "hos_linear_dgrad", ptr(dY), dY.stride(0), wptr, W.stride(0), Npad, mptr,
        0 if mask_src is None else mask_src.stride(0), ptr(out), out.stride(0), M, K, int(accumulate)))
    return out


def linear_wgrad(dY: torch.Tensor, X: torch.Tensor, dW: torch.Tensor, db: Optional[torch.Tensor], N: int, K: int,
                 w_col0: int = 0, splits: int = 0):
    """dW[:N, w_col0:w_col0+K] += dY[:, :N]^T @ X[:, :K];  db[:N] += colsum(dY[:, :N])."""
    M = dY.shape[0]
    if WGRAD_TR and 128 < N <= 256 and M >= 16384 and _lib.load().hos_get_gemm_mode() == GEMM_BF16X3:
        # many rows, thin layer: staged-planes kernel with transposed LDS reads (hos_mlpbwd.hip), K in chunks of <= 256
        for k0 in range(0, K, 256):
            kc = min(256, K - k0)
            ws = _bwd_workspace(dY.device, M, N, kc, False)
            _timed(f"wgrad_tr[M={N},N={kc},K={M}]", 2.0 * M * N * kc, lambda: call(
                "hos_linear_wgrad_tr", ptr(dY), dY.stride(0), ptr(X) + 4 * k0, X.stride(0), ptr(dW) + 4 * (w_col0 + k0), dW.stride(0),
                ptr(db) if k0 == 0 else None, M, N, kc, ptr(ws), ws.numel()))
        return
    _timed(f"gemm_wgrad[M={N},N={K},K={M}]", 2.0 * M * N * K, lambda: call(
        "hos_linear_wgrad", ptr(dY), dY.stride(0), ptr(X), X.stride(0), ptr(dW) + 4 * w_col0, dW.stride(0),
        ptr(db), M, N, K, splits))


def linear_bwd_fused(dY: torch.Tensor, X: torch.Tensor, W: torch.Tensor, dW: torch.Tensor, db: Optional[torch.Tensor], N: int,
                     K: int, out: Optional[torch.Tensor], relu_mask: bool, w_col0: int = 0, rows_dev: Optional[torch.Tensor] = None):
    """One thin layer's whole backward in one pass over (dY, X) (hos_mlpbwd.hip; N, K <= 128):
    out[M,:K] = (dY[:, :N] @ W[:N, w_col0:w_col0+K]) * (X > 0 if relu_mask);  dW[:N, w_col0:..+K] += dY^T @ X;  db += colsum(dY)."""
    M = dY.shape[0]
    ws = _bwd_workspace(dY.device, M, N, K, True)
    _timed(f"mlp_bwd_fused[M={M},N={N},K={K}]", 4.0 * M * N * K, lambda: call(
        "hos_linear_bwd_fused", ptr(dY), dY.stride(0), ptr(X), X.stride(0), ptr(W) + 4 * w_col0, W.stride(0),
        ptr(out), 0 if out is None else out.stride(0), ptr(dW) + 4 * w_col0, dW.stride(0), ptr(db), M, N, K, int(relu_mask),
        ptr(ws), ws.numel(), ptr(rows_dev, torch.int32)))
    return out


_BWD_WS = {}
_BWD_DEFER = {"on": False, "offset": 0}


def _stream_key(device):
    """Scratch buffers that a launch and its follow-up reduce share are only ordered within ONE stream: key them by stream."""
    d = torch.device(device)
    return (str(d), torch.cuda.current_stream(d).cuda_stream if d.type == "cuda" else 0)


def _bwd_workspace(device, M: int = 0, N: int = 0, K: int = 0, fused: bool = False, need: Optional[int] = None) -> torch.Tensor:
    """Per-workgroup dW / db partials of hos_linear_bwd_fused / hos_linear_wgrad_tr: 256 slabs of up to 256 x 256 + 256 floats
    (67 MB) per call, one buffer per device -- launches on a stream are ordered and the reduce kernel that reads a call's slabs
    is enqueued by the same call.  Inside `deferred_bwd_reduce()` the reductions are postponed to one batched launch, so every
    call gets its OWN region of the (then larger) buffer."""
    key = _stream_key(device)            # one buffer per (device, stream): the two branches of a stage-3 step run on two streams
    if not _BWD_DEFER["on"]:
        if key not in _BWD_WS:
            _BWD_WS[key] = torch.empty(256 * (256 * 256 + 256), device=device)
        return _BWD_WS[key][:256 * (256 * 256 + 256)]
    if need is None:
        need = int(_lib.load().hos_mlp_bwd_ws_floats(M, N, K, int(fused)))
    need = max((need + 3) // 4 * 4, 4)
    ws = _BWD_WS.get(key)
    if ws is None or ws.numel() < max(need, BWD_DEFER_WS_FLOATS):
        _BWD_WS[key] = ws = torch.empty(max(need, BWD_DEFER_WS_FLOATS), device=device)      # (replaces the 67 MB buffer of the immediate mode)
    if _BWD_DEFER["offset"] + need > ws.numel():      # full: run the recorded reductions now and start over at the front
        call("hos_mlp_bwd_flush")
        _BWD_DEFER["offset"] = 0
    off = _BWD_DEFER["offset"]
    _BWD_DEFER["offset"] = off + need
    return ws[off:off + need]


BWD_DEFER = os.environ.get("HOS_DEFER_REDUCE", "1") != "0"
BWD_DEFER_WS_FLOATS = 160 * 1024 * 1024       # 640 MB: the eight 256-wide layers of the canonical MLP (67 MB of slabs each) in one batch


class deferred_bwd_reduce:
    """with deferred_bwd_reduce(): the slab reductions of the linear_bwd_fused / linear_wgrad(tr) calls inside run as ONE
    batched launch at the exit (hos_mlp_bwd_defer / hos_mlp_bwd_flush); dW / db are complete after the block."""

    def __enter__(self):
        self.nested = _BWD_DEFER["on"] or not BWD_DEFER
        if not self.nested:
            _BWD_DEFER["on"], _BWD_DEFER["offset"] = True, 0
            _lib.check(_lib.load().hos_mlp_bwd_defer(1), "hos_mlp_bwd_defer")
        return self

    def __exit__(self, *exc):
        if not self.nested:
            _BWD_DEFER["on"] = False
            _lib.check(_lib.load().hos_mlp_bwd_defer(0), "hos_mlp_bwd_defer")
            call("hos_mlp_bwd_flush")
        return False


FUSED_THIN_BWD = os.environ.get("HOS_FUSED_BWD", "1") != "0"
WGRAD_TR = os.environ.get("HOS_WGRAD_TR", "1") != "0"
ROWDOT_HEADS = os.environ.get("HOS_ROWDOT_HEADS", "1") != "0"      # one-column heads of the planes MLPs as a row dot (hos_planes_rowdot)
THIN_GEMM = os.environ.get("HOS_THIN_GEMM", "1") != "0"
WGRAD_WS = os.environ.get("HOS_WGRAD_WS", "1") == "1"       # planes WGRAD: split-K partials through the slab workspace
WGRAD_WS_MIN = int(os.environ.get("HOS_WGRAD_WS_MIN", "0"))  # ... for gradients of at least this many elements


# ------------------------------------------------------------------------------------------ fused MLP chain (hos_chain.hip)
MLP_CHAIN = os.environ.get("HOS_MLP_CHAIN", "1") != "0"
MLP_CHAIN_MIN_ROWS = int(os.environ.get("HOS_MLP_CHAIN_MIN_ROWS", "4096"))
# the non-rigid chain with the per-frame condition code folded into the first layer's bias (hos_chain.hip, FOLD)
MLP_CHAIN_FOLD = os.environ.get("HOS_CHAIN_FOLD", "1") != "0"
_FOLD_WS = {}


def fold_grad_workspace(device) -> torch.Tensor:
    """[128 * 64 + 128] floats per (device, stream): gradient buffers of a folded first layer between its backward launch and
    hos_mlp_chain_unfold_grad (both enqueued by the same chain backward, so stream order is the only synchronisation needed)."""
    key = _stream_key(device)
    if key not in _FOLD_WS:
        _FOLD_WS[key] = torch.empty(128 * 64 + 128, device=device)
    return _FOLD_WS[key]
# ---- backward of the non-rigid MLP as three group launches, dZ on chip between the layers of a group (hos_mlpbwd.hip, chain_bwd_kernel)
MLP_CHAIN_BWD = os.environ.get("HOS_CHAIN_BWD", "1") != "0"
MLP_CHAIN_BWD_MIN_ROWS = int(os.environ.get("HOS_CHAIN_BWD_MIN_ROWS", "16384"))
_CB_IMAGES = {}
_CB_IMAGES_MAX = 16        # (key, groups, device, stream) entries kept; oldest evicted first (buffers are repacked every backward)


def _int_array(vals):
    import ctypes
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def mlp_chain_bwd_images(key, cfgs, device):
    """One int16 buffer per (cfg, step) of the groups `cfgs`, cached per (key, device, stream)."""
    lib = _lib.load()
    k = (key, tuple(cfgs)) + _stream_key(device)
    bufs = _CB_IMAGES.get(k)
    if bufs is None:
        while len(_CB_IMAGES) >= _CB_IMAGES_MAX:        # a process that rebuilds Network objects must not accumulate images
            _CB_IMAGES.pop(next(iter(_CB_IMAGES)))
        bufs = _CB_IMAGES[k] = [[torch.empty(int(lib.hos_mlp_chain_bwd_image_bytes(c, s)) // 2, dtype=torch.int16, device=device)
                                 for s in range(int(lib.hos_mlp_chain_bwd_steps(c)))] for c in cfgs]
    return bufs


def mlp_chain_bwd_pack(jobs):
    """jobs: [(cfg, step, W [rows, ld] fp32 view, w_col0, N, K, image)], at most 8 per launch."""
    import ctypes
    for i in range(0, len(jobs), 8):
        js = jobs[i:i + 8]
        Wp = (ctypes.c_void_p * len(js))(*[ptr(j[2]) + 4 * j[3] for j in js])
        call("hos_mlp_chain_bwd_pack", len(js), _int_array([j[0] for j in js]), _int_array([j[1] for j in js]), Wp,
             _int_array([j[2].stride(0) for j in js]), _int_array([j[4] for j in js]), _int_array([j[5] for j in js]),
             (ctypes.c_void_p * len(js))(*[ptr(j[6], torch.int16) for j in js]))


def mlp_chain_bwd(cfg, dZ, X, images, dXout, dW, w_col0, db, N, K, rows_dev=None):
    """One group of layer steps (include/hosrender.h: hos_mlp_chain_bwd).  Per step: X[s] input rows, images[s] packed weight,
    dXout[s] (tensor for '-> HBM' steps, else None), dW[s] (+ w_col0[s]) / db[s] gradient views, N[s], K[s]."""
    import ctypes
    M = dZ.shape[0]
    S = len(X)
    need = int(_lib.load().hos_mlp_chain_bwd_ws_floats(cfg, M))
    ws = _bwd_workspace(dZ.device, need=need)
    flop = sum(4.0 * M * n * k for n, k in zip(N, K))
    _timed(f"mlp_chain_bwd{cfg}[M={M}]", flop, lambda: call(
        "hos_mlp_chain_bwd", cfg, ptr(dZ), dZ.stride(0), M, ptr(rows_dev, torch.int32), _ptr_array(list(X)), _int_array([x.stride(0) for x in X]),
        (ctypes.c_void_p * S)(*[ptr(im, torch.int16) for im in images]), _ptr_array(list(dXout)),
        _int_array([0 if o is None else o.stride(0) for o in dXout]),
        (ctypes.c_void_p * S)(*[ptr(g) + 4 * c for g, c in zip(dW, w_col0)]), _int_array([g.stride(0) for g in dW]),
        _ptr_array(list(db)), _int_array(N), _int_array(K), ptr(ws), ws.numel()))


def mlp_chain_buffers(device):
    """(chain planes [bytes/2] int16, aux [floats]) for one 6 x 128 MLP -- filled by mlp_chain_pack."""
    lib = _lib.load()
    return (torch.empty(int(lib.hos_mlp_chain_weight_bytes()) // 2, dtype=torch.int16, device=device),
            torch.empty(int(lib.hos_mlp_chain_aux_floats()), device=device))


def mlp_chain_pack(weights, biases, planes, aux):
    """weights: 7 fp32 matrices (views [rows, ld] of the flat parameter buffer, nn.Linear layout), biases: 7 vectors."""
    import ctypes
    ldw = (ctypes.c_int * 7)(*[int(w.stride(0)) for w in weights])
    call("hos_mlp_chain_pack", _ptr_array(list(weights)), ldw, _ptr_array(list(biases)), ptr(planes, torch.int16), ptr(aux))


def mlp_chain_pack_fold(weights, biases, cond, nfeat, planes, aux, w0h):
    """As mlp_chain_pack with the condition code `cond` (the same in every row of the coming launches) folded into the first
    layer's bias; `w0h` [128, 64] receives the aligned fp32 copy of W0's feature columns (the backward pass's operand)."""
    import ctypes
    ldw = (ctypes.c_int * 7)(*[int(w.stride(0)) for w in weights])
    call("hos_mlp_chain_pack_fold", _ptr_array(list(weights)), ldw, _ptr_array(list(biases)), ptr(cond), cond.numel(), nfeat,
         ptr(planes, torch.int16), ptr(aux), ptr(w0h))


def mlp_chain_unfold_grad(gw0h, db, cond, nfeat, gW0, gb0):
    """The folded first layer's gradients back into the reference-shaped ones: gW0[:, C:] += gw0h, gW0[:, :C] += db (x) cond, gb0 += db."""
    call("hos_mlp_chain_unfold_grad", ptr(gw0h), ptr(db), ptr(cond), cond.numel(), nfeat, ptr(gW0), gW0.stride(0), ptr(gb0))


def mlp_chain128_fwd(E, PE, x, planes, aux, acts, xyz, rows_dev=None):
    """xyz = x + MLP(E | PE) with the six hidden activations written once into `acts` (mlp_offset.py:54-70 in one launch).
    E = None: the folded form (planes / aux from mlp_chain_pack_fold), PE is the only per-row operand."""
    P = x.shape[0]
    flop = 101120 if E is not None else 101120 - 128 * 64
    _timed(f"mlp_chain128{'' if E is not None else 'f'}[M={P}]", 2.0 * P * flop, lambda: call(
        "hos_mlp_chain128_fwd", ptr(E), 0 if E is None else E.stride(0), ptr(PE), PE.stride(0), ptr(x), ptr(planes, torch.int16), ptr(aux),
        _ptr_array(list(acts)), acts[0].stride(0), ptr(xyz), P, ptr(rows_dev, torch.int32)))


# ------------------------------------------------------------------------------------------ fp16 range guard
_RANGE_FLAG = {}
RANGE_GUARD = os.environ.get("HOS_RANGE_GUARD", "1") != "0"


def arm_range_flag(device) -> torch.Tensor:
    """Register a device word with the library (hos_set_range_flag): forward epilogues set it to 1 when a hidden activation
    exceeds the range the fp16 (hi, lo) operand format represents exactly (|x| <= 6e4 flagged, 65 504 exact, 131 008 hard limit)."""
    key = str(device)
    if key not in _RANGE_FLAG:
        _RANGE_FLAG[key] = torch.zeros(1, dtype=torch.int32, device=device)
    _lib.check(_lib.load().hos_set_range_flag(ptr(_RANGE_FLAG[key], torch.int32)), "hos_set_range_flag")
    return _RANGE_FLAG[key]


def range_events(device, reset: bool = True) -> int:
    """1 if any forward GEMM since the last reset produced an activation outside the exact fp16 hi/lo range (host sync)."""
    flag = _RANGE_FLAG.get(str(device))
    if flag is None:
        return 0
    v = int(flag.item())
    if v and reset:
        flag.zero_()
    return v


_RANGE_SKIPS = {}
_RANGE_SKIPS_SEEN = {}


def range_guard_words(device):
    """(guard word, {skipped-steps counter, ticket scratch}) for hos_adam_multi, or (None, None) when the guard is off.  The word is
    the library's range flag (registered here if it was not yet): the forward epilogues of a training step OR it, the optimiser
    launch of the same step reads it, skips the update if it is set, counts the skip and clears the word for the next step (the
    kernel's last workgroup does; round 5).  Never allocates while a graph is being captured."""
    if not RANGE_GUARD:
        return None, None
    key = str(device)
    if key not in _RANGE_FLAG or key not in _RANGE_SKIPS:
        if torch.cuda.is_current_stream_capturing():
            return None, None
        arm_range_flag(device)
        _RANGE_SKIPS[key] = torch.zeros(2, dtype=torch.int32, device=device)
    return _RANGE_FLAG[key], _RANGE_SKIPS[key]


def range_skips(device, since_last_poll: bool = False) -> int:
    """Optimiser steps the device-side range guard has skipped (one 4-byte read: poll it rarely); since_last_poll: only the new ones."""
    key = str(device)
    t = _RANGE_SKIPS.get(key)
    if t is None:
        return 0
    n = int(t[0].item())
    if since_last_poll:
        n, _RANGE_SKIPS_SEEN[key] = n - _RANGE_SKIPS_SEEN.get(key, 0), n
    return n


def guarded_forward(module, device, run):
    """Run `run()` (a no-grad forward of `module`); if the fp16 range flag fires, switch the module to exact-fp32 MFMA for good
    (`module.gemm_mode = GEMM_FP32`) and run it again.  Costs one 4-byte device read per call; only used without autograd
    (evaluation / inference), training loops poll `range_events` every few hundred steps instead (train.check_range)."""
    mode = getattr(module, "gemm_mode", None)
    if mode is not None:
        with gemm_mode(mode):
            return run()
    if not RANGE_GUARD or torch.is_grad_enabled() or torch.cuda.is_current_stream_capturing():
        return run()
    arm_range_flag(device)
    out = run()
    if range_events(device):
        import warnings
        warnings.warn(f"{type(module).__name__}: hidden activations beyond the exact fp16 hi/lo range (|x| > 6e4); "
                      "switching this module to exact fp32 MFMA")
        module.gemm_mode = GEMM_FP32
        with gemm_mode(GEMM_FP32):
            out = run()
    return out


class gemm_mode:
    """Context manager: run the GEMM launches THIS THREAD makes inside the block in another arithmetic mode (the mode is read
    when a kernel is launched, so this also works while a graph is being captured).  A per-thread override in the library
    (hos_set_thread_gemm_mode) and here: other host threads -- the autograd engine's worker, another module pinned to another
    mode -- keep theirs."""

    def __init__(self, mode: int):
        self.mode = mode

    def __enter__(self):
        self.prev = getattr(_tls, "mode", None)
        _set_thread_mode(self.mode)

    def __exit__(self, *exc):
        _set_thread_mode(self.prev)
        return False


# ------------------------------------------------------------------------------------------ ConvTranspose3d(4, 2, 1)
class _Deconv3d(torch.autograd.Function):
    """ConvTranspose3d(kernel 4, stride 2, padding 1), batch 1, channel-last activations, as
    GEMM (exact fp32 MFMA) + gather (hos_deconv3d_*).  x [D^3, Cin], weight [Cin, Cout, 4,4,4] (the reference's
    parameter layout), bias [Cout] -> [(2D)^3, Cout], optionally followed by LeakyReLU(0.2).
    If `weight.grad` / `bias.grad` alias a flat gradient buffer the parameter gradients are accumulated in place."""

    @staticmethod
    def forward(ctx, x, weight, bias, D, leaky):
        Cin, Cout = weight.shape[0], weight.shape[1]
        M = D * D * D
        x = x.contiguous()
        Wm = weight.detach().view(Cin, Cout * 64)
        ycol = torch.empty(M, Cout * 64, device=x.device)
        with gemm_mode(GEMM_FP32):
            linear_dgrad(x, Wm, Cin, Cout * 64, ycol)
        out = torch.empty(8 * M, Cout, device=x.device)
        call("hos_deconv3d_col2im", ptr(ycol), ptr(bias.detach()), D, Cout, 0.2, int(leaky), ptr(out))
        ctx.save_for_backward(x, weight, bias, out)
        ctx.D, ctx.leaky = D, leaky
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight, bias, out = ctx.saved_tensors
        D, leaky = ctx.D, ctx.leaky
        Cin, Cout = weight.shape[0], weight.shape[1]
        M = D * D * D
        g = g.contiguous()
        in_place = weight.grad is not None and weight.grad.is_contiguous()
        db_in_place = in_place and bias.grad is not None and bias.grad.is_contiguous()
        db = bias.grad if db_in_place else zeros(Cout, g.device)
        # LeakyReLU backward + bias gradient: one pass (was where / mul / sum / add: 4 torch launches per layer, the [32768, 27]
        # column sum alone 88 us)
        dpre = torch.empty_like(g) if leaky else g
        call("hos_deconv3d_dpre", ptr(g), ptr(out), 8 * M, Cout, 0.2, int(leaky), ptr(dpre) if leaky else None, ptr(db))
        dycol = torch.empty(M, Cout * 64, device=g.device)
        call("hos_deconv3d_im2col", ptr(dpre), D, Cout, ptr(dycol))
        Wm = weight.detach().view(Cin, Cout * 64)
        dx = None
        if ctx.needs_input_grad[0]:
            # [M <= 4096, Cin <= 1024] from a reduction of Cout*64 = 1 728 .. 32 768: a tiny output with a huge reduction --
            # the forward-form tile kernel with the reduction split over ~512 workgroups (hos_linear_fwd_splitk; the library
            # GEMM picked for these shapes took 37-107 us per layer)
            dx = torch.empty(M, Cin, device=g.device)
            call("hos_linear_fwd_splitk", ptr(dycol), dycol.stride(0), ptr(Wm), Wm.stride(0), ptr(dx), dx.stride(0), M, Cin, Cout * 64)
        with gemm_mode(GEMM_FP32):
            gW = weight.grad.view(Cin, Cout * 64) if in_place else zeros((Cin, Cout * 64), g.device)
            if M <= 8:       # a few voxels against 33-134 MB of weights: outer-product stream, not a tiled GEMM (at 64 voxels the
                             # tiled GEMM wins: 28 vs 67 us, scripts/bench_decoder.py)
                call("hos_outer_accum", ptr(x), x.stride(0), ptr(dycol), dycol.stride(0), ptr(gW), gW.stride(0), M, Cin, Cout * 64)
            else:
                linear_wgrad(x, dycol, gW, None, Cin, Cout * 64)
        if db_in_place:
            return dx, (None if in_place else gW.view_as(weight)), None, None, None
        return dx, (None if in_place else gW.view_as(weight)), db, None, None


def deconv3d(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, D: int, leaky: bool) -> torch.Tensor:
    return _Deconv3d.apply(x, weight, bias, D, leaky)


class _Deconv3dFirst(torch.autograd.Function):
    """ConvTranspose3d(4, 2, 1) on ONE input voxel: only the 2 x 2 x 2 centre taps of the kernel reach the 2^3 output, so the
    layer is x [1, Cin] @ Wc [Cin, 8 * Cout] with the live taps stored tap-major (human_nerf.Network._w0c): the product, viewed
    as [8, Cout], IS the channel-last output.  `wc` / `gwc` are views of the flat parameter / gradient buffers (the weight
    gradient is accumulated in place, like every other HIP weight gradient); exact fp32 MFMA like the other decoder layers."""

    @staticmethod
    def forward(ctx, x, wc, gwc, bias, leaky):
        Cin, N8 = wc.shape
        Cout = N8 // 8
        x = x.contiguous()
        # one row against 16.8 MB of weights: the fixed-order two-launch GEMV with bias + LeakyReLU in its second launch
        # (hos_gemv_rowvec; the 32-row GEMM tile ran this on 32 workgroups: 34 us + two torch launches)
        out = torch.empty(8, Cout, device=x.device)
        ws = torch.empty(int(_lib.load().hos_gemv_ws_floats(Cin, N8)), device=x.device)
        call("hos_gemv_rowvec", ptr(x), ptr(wc), wc.stride(0), Cin, N8, ptr(bias.detach()), Cout, 0.2, int(leaky), ptr(ws), ptr(out))
        ctx.save_for_backward(x, bias, out)
        ctx.wc, ctx.gwc, ctx.leaky = wc, gwc, leaky
        return out

    @staticmethod
    def backward(ctx, g):
        x, bias, out = ctx.saved_tensors
        wc, gwc = ctx.wc, ctx.gwc
        Cin, N8 = wc.shape
        g = g.contiguous()
        Cout = N8 // 8
        db_in_place = bias.grad is not None and bias.grad.is_contiguous()
        db = None if db_in_place else zeros(Cout, g.device)
        dpre = torch.empty_like(g) if ctx.leaky else g
        call("hos_deconv3d_dpre", ptr(g), ptr(out), 8, Cout, 0.2, int(ctx.leaky), ptr(dpre) if ctx.leaky else None,
             ptr(bias.grad if db_in_place else db))
        dycol = dpre.view(1, N8)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(1, Cin, device=g.device)
            call("hos_linear_fwd_splitk", ptr(dycol), dycol.stride(0), ptr(wc), wc.stride(0), ptr(dx), dx.stride(0), 1, Cin, N8)
        call("hos_outer_accum", ptr(x), x.stride(0), ptr(dycol), dycol.stride(0), ptr(gwc), gwc.stride(0), 1, Cin, N8)
        return dx, None, None, db, None


def deconv3d_first(x: torch.Tensor, wc: torch.Tensor, gwc: torch.Tensor, bias: torch.Tensor, leaky: bool) -> torch.Tensor:
    return _Deconv3dFirst.apply(x, wc, gwc, bias, leaky)


# ---- round 5: the same two layers with the weight SHARDED over the data-parallel ranks by input channel -------------------------
# The volume decoder sees no ray (its input is a learned constant), so under DDP every rank used to stream the same 136 MB of
# weights forward, the same again twice backward and 253 MB of Adam state -- ~1.1 ms of a 6 ms 512-ray step that does not shrink
# with the ray count.  Rank r owns rows [c0, c1) of a layer's [Cin, Cout * 64] weight (a CONTIGUOUS range of the flat buffer, so
# the optimiser spans stay ranges): forward  = partial product of its input columns, col2im (linear), SUM over the ranks, bias +
# LeakyReLU;  backward = bias gradient / LeakyReLU mask / im2col on the full (already rank-summed) output gradient (replicated,
# small), its rows of the weight gradient (complete: no reduction), its columns of the input gradient, all-gathered.  Per
# sharded layer and step: one all-reduce of [(2D)^3, Cout] and one all-gather of [D^3, Cin] floats (<= 512 KB).  `comm` is a
# train.ShardComm (RCCL through libhoscomm inside a captured graph, or torch.distributed).
class _Deconv3dSharded(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, D, leaky, comm):
        Cin, Cout = weight.shape[0], weight.shape[1]
        cs = Cin // comm.world
        c0 = comm.rank * cs
        M = D * D * D
        x = x.contiguous()
        Wm = weight.detach().view(Cin, Cout * 64)
        ycol = torch.empty(M, Cout * 64, device=x.device)
        with gemm_mode(GEMM_FP32):          # ycol = x[:, c0:c1] @ Wm[c0:c1]  (the column slice of x by pointer + row stride)
            call("hos_linear_dgrad", ptr(x) + 4 * c0, x.stride(0), ptr(Wm[c0:c0 + cs]), Wm.stride(0), cs, None, 0,
                 ptr(ycol), ycol.stride(0), M, Cout * 64, 0)
        out = torch.empty(8 * M, Cout, device=x.device)
        call("hos_deconv3d_col2im", ptr(ycol), None, D, Cout, 0.2, 0, ptr(out))
        comm.all_reduce_sum(out)
        call("hos_bias_lrelu", ptr(out), ptr(bias.detach()), 8 * M, Cout, 0.2, int(leaky))
        ctx.save_for_backward(x, weight, bias, out)
        ctx.D, ctx.leaky, ctx.comm = D, leaky, comm
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight, bias, out = ctx.saved_tensors
        D, leaky, comm = ctx.D, ctx.leaky, ctx.comm
        Cin, Cout = weight.shape[0], weight.shape[1]
        cs = Cin // comm.world
        c0 = comm.rank * cs
        M = D * D * D
        g = g.contiguous()
        if not (weight.grad is not None and weight.grad.is_contiguous() and bias.grad is not None and bias.grad.is_contiguous()):
            raise _lib.HosLibraryError("sharded decoder layers accumulate into the flat gradient buffer (FlatStore-bound parameters)")
        dpre = torch.empty_like(g) if leaky else g
        call("hos_deconv3d_dpre", ptr(g), ptr(out), 8 * M, Cout, 0.2, int(leaky), ptr(dpre) if leaky else None, ptr(bias.grad))
        dycol = torch.empty(M, Cout * 64, device=g.device)
        call("hos_deconv3d_im2col", ptr(dpre), D, Cout, ptr(dycol))
        Wm = weight.detach().view(Cin, Cout * 64)
        dx = None
        if ctx.needs_input_grad[0]:
            mine = torch.empty(M, cs, device=g.device)
            call("hos_linear_fwd_splitk", ptr(dycol), dycol.stride(0), ptr(Wm[c0:c0 + cs]), Wm.stride(0), ptr(mine), cs, M, cs, Cout * 64)
            parts = comm.all_gather(mine)                       # [world, M, cs]
            dx = torch.empty(M, Cin, device=g.device)
            call("hos_shard_interleave", ptr(parts), comm.world, M, cs, ptr(dx))
        gW = weight.grad.view(Cin, Cout * 64)
        with gemm_mode(GEMM_FP32):
            if M <= 8:
                call("hos_outer_accum", ptr(x) + 4 * c0, x.stride(0), ptr(dycol), dycol.stride(0), ptr(gW[c0:c0 + cs]), gW.stride(0), M, cs, Cout * 64)
            else:                           # gW[c0:c1] += x[:, c0:c1]^T @ dycol
                call("hos_linear_wgrad", ptr(x) + 4 * c0, x.stride(0), ptr(dycol), dycol.stride(0), ptr(gW[c0:c0 + cs]), gW.stride(0),
                     None, M, cs, Cout * 64, 0)
        return dx, None, None, None, None, None


def deconv3d_sharded(x, weight, bias, D: int, leaky: bool, comm) -> torch.Tensor:
    return _Deconv3dSharded.apply(x, weight, bias, D, leaky, comm)


class _Deconv3dFirstSharded(torch.autograd.Function):
    """`_Deconv3dFirst` (one input voxel, live taps [Cin, 8 * Cout]) with the rows [c0, c1) of this rank."""

    @staticmethod
    def forward(ctx, x, wc, gwc, bias, leaky, comm):
        Cin, N8 = wc.shape
        Cout = N8 // 8
        cs = Cin // comm.world
        c0 = comm.rank * cs
        x = x.contiguous()
        out = torch.empty(8, Cout, device=x.device)
        ws = torch.empty(int(_lib.load().hos_gemv_ws_floats(cs, N8)), device=x.device)
        call("hos_gemv_rowvec", ptr(x) + 4 * c0, ptr(wc[c0:c0 + cs]), wc.stride(0), cs, N8, None, Cout, 0.2, 0, ptr(ws), ptr(out))
        comm.all_reduce_sum(out)
        call("hos_bias_lrelu", ptr(out), ptr(bias.detach()), 8, Cout, 0.2, int(leaky))
        ctx.save_for_backward(x, bias, out)
        ctx.wc, ctx.gwc, ctx.leaky, ctx.comm = wc, gwc, leaky, comm
        return out

    @staticmethod
    def backward(ctx, g):
        x, bias, out = ctx.saved_tensors
        wc, gwc, comm = ctx.wc, ctx.gwc, ctx.comm
        Cin, N8 = wc.shape
        Cout = N8 // 8
        cs = Cin // comm.world
        c0 = comm.rank * cs
        g = g.contiguous()
        if not (bias.grad is not None and bias.grad.is_contiguous()):
            raise _lib.HosLibraryError("sharded decoder layers accumulate into the flat gradient buffer (FlatStore-bound parameters)")
        dpre = torch.empty_like(g) if ctx.leaky else g
        call("hos_deconv3d_dpre", ptr(g), ptr(out), 8, Cout, 0.2, int(ctx.leaky), ptr(dpre) if ctx.leaky else None, ptr(bias.grad))
        dycol = dpre.view(1, N8)
        dx = None
        if ctx.needs_input_grad[0]:
            mine = torch.empty(1, cs, device=g.device)
            call("hos_linear_fwd_splitk", ptr(dycol), dycol.stride(0), ptr(wc[c0:c0 + cs]), wc.stride(0), ptr(mine), cs, 1, cs, N8)
            dx = comm.all_gather(mine).reshape(1, Cin)           # one row: rank order IS channel order
        call("hos_outer_accum", ptr(x) + 4 * c0, x.stride(0), ptr(dycol), dycol.stride(0), ptr(gwc[c0:c0 + cs]), gwc.stride(0), 1, cs, N8)
        return dx, None, None, None, None, None


def deconv3d_first_sharded(x, wc, gwc, bias, leaky: bool, comm) -> torch.Tensor:
    return _Deconv3dFirstSharded.apply(x, wc, gwc, bias, leaky, comm)


DECODER_HEAD_INPLACE = True      # the training step's loss.backward(); set False around torch.autograd.grad / gradcheck of this node


class _DecoderHead(torch.autograd.Function):
    """h [1, N] = LeakyReLU(0.2)(W [N, K] . e [K] + b): `block_mlp` of the volume decoder on its constant embedding
    (network_util.py:21-30, deconv_vol_decoder.py:36-37) -- one row-dot launch (was F.linear -> a library GEMM, + leaky_relu);
    the backward accumulates into the parameters' flat gradients in place when they alias them."""

    @staticmethod
    def forward(ctx, emb, weight, bias):
        N, K = weight.shape
        y = torch.empty(1, N, device=emb.device)
        call("hos_rowdot_lrelu_fwd", ptr(emb.detach().contiguous()), ptr(weight.detach()), weight.stride(0), ptr(bias.detach()), N, K, 0.2, ptr(y))
        ctx.save_for_backward(emb, weight, bias, y)
        return y

    @staticmethod
    def backward(ctx, g):
        emb, weight, bias, y = ctx.saved_tensors
        N, K = weight.shape
        g = g.contiguous()
        # in place only when the flat store owns these gradients (FlatStore marks its aliased `.grad`s): torch.autograd.grad(...),
        # gradient checks or a second backward over the graph then get (gx, gW, gb) back like from any other node
        ok = lambda p_: p_.grad is not None and p_.grad.is_contiguous() and getattr(p_, "_hos_flat_grad", False)
        in_place = ok(emb) and ok(weight) and ok(bias) and DECODER_HEAD_INPLACE
        if in_place:
            gW, gb, gx = weight.grad, bias.grad, emb.grad
        else:
            gW, gb, gx = zeros_many([(N, K), (N,), (K,)], g.device)
        call("hos_rowdot_lrelu_bwd", ptr(g), ptr(y), ptr(emb.detach().contiguous()), ptr(weight.detach()), weight.stride(0), N, K, 0.2,
             ptr(gW), gW.stride(0), ptr(gb), ptr(gx))
        return (None, None, None) if in_place else (gx, gW, gb)


def decoder_head(emb, weight, bias):
    return _DecoderHead.apply(emb, weight, bias)


class _VolumeSoftmax(torch.autograd.Function):
    """vol [C, V, V, V] = softmax_c(z [V^3, C] + log prior) (deconv_vol_decoder.py:38-42): one launch each way (was log, add,
    softmax on a transposed view + their three backward launches and a clone)."""

    @staticmethod
    def forward(ctx, z, prior):
        V3, C = z.shape
        V = round(V3 ** (1.0 / 3.0))
        z = z.contiguous()
        vol = torch.empty(C, V, V, V, device=z.device)
        call("hos_volume_softmax_fwd", ptr(z), ptr(prior.detach().contiguous().float()), C, V3, ptr(vol))
        ctx.save_for_backward(vol)
        return vol

    @staticmethod
    def backward(ctx, g):
        vol, = ctx.saved_tensors
        C = vol.shape[0]
        V3 = vol.numel() // C
        gz = torch.empty(V3, C, device=vol.device)
        g = g.contiguous()
        call("hos_volume_softmax_bwd", ptr(g), ptr(vol), C, V3, ptr(gz))
        return gz, None


def volume_softmax(z, prior):
    return _VolumeSoftmax.apply(z, prior)


class _VolumePair(torch.autograd.Function):
    """vol [C, V, V, V] -> (vol, vol_cl [V, V, V, 32]): the channel-major volume the backward warp samples per bone and the
    channel-last copy of its first K bone channels for the forward warp (all bones at one position).  The backward forms the
    volume's gradient from both consumers in one launch (was slice + permute + pad, their backward and an accumulation add)."""

    @staticmethod
    def forward(ctx, vol, K):
        C, V = vol.shape[0], vol.shape[-1]
        vol = vol.contiguous()
        cl = torch.empty(V, V, V, 32, device=vol.device)
        call("hos_volume_channel_last", ptr(vol), K, V * V * V, ptr(cl))
        ctx.dims = (C, K, V)
        ctx.set_materialize_grads(False)
        return vol.view_as(vol), cl

    @staticmethod
    def backward(ctx, g_vol, g_cl):
        if g_vol is None and g_cl is None:
            return None, None
        C, K, V = ctx.dims
        ref = g_vol if g_vol is not None else g_cl
        g = torch.empty(C, V, V, V, device=ref.device)
        g_vol = None if g_vol is None else g_vol.contiguous()
        g_cl = None if g_cl is None else g_cl.contiguous()
        call("hos_volume_pair_bwd", ptr(g_vol), ptr(g_cl), C, K, V * V * V, ptr(g))
        return g, None


def volume_pair(vol, K: int):
    return _VolumePair.apply(vol, K)


# ------------------------------------------------------------------------------------------ rays
_U_CACHE = {}


def _u_base(S: int, randomized: bool, device) -> Tuple[torch.Tensor, float]:
    """The torch.linspace grid of H:354 / H:363 (built with torch on the host so it is bit-identical)."""
    key = (S, randomized, str(device))
    if key not in _U_CACHE:
        if randomized:
            u_max = EPS + (1 - EPS) / S
            max_jitter = (1 - u_max) / (S - 1) - EPS
            u = torch.linspace(0, 1 - u_max, S)
        else:
            pad = 1 / (2 * S)
            max_jitter = 0.0
            u = torch.linspace(pad, 1 - pad - EPS, S)
        _U_CACHE[key] = (u.to(device), float(np.float32(max_jitter)))
    return _U_CACHE[key]


def resample(sdist_prev: torch.Tensor, w_prev: torch.Tensor, S: int, dilation: float, anneal: float,
             randomized: bool, near: float, far: float, jitter: Optional[torch.Tensor] = None,
             resample_padding: float = 0.0, want_index: bool = False, train_frac_dev: Optional[torch.Tensor] = None,
             anneal_slope: float = 10.0):
    """Fused max_dilate -> logits -> inverse-CDF -> interval edges -> s_to_t.  Returns (sdist, tdist[, idx]).
    `train_frac_dev` (1-element device tensor): the anneal factor is evaluated on the device from it (graph replay)."""
    B, n = w_prev.shape
    dev = w_prev.device
    u, scale = _u_base(S, randomized, dev)
    if randomized and jitter is None:
        jitter = torch.rand(B, device=dev)
    if not randomized:
        jitter = None
    sdist = torch.empty(B, S + 1, device=dev)
    tdist = torch.empty(B, S + 1, device=dev)
    idx = torch.empty(B, S, dtype=torch.int32, device=dev) if want_index else None
    call("hos_resample", ptr(sdist_prev.detach()), ptr(w_prev.detach()), n, B, S, float(dilation), float(anneal),
         ptr(train_frac_dev), float(anneal_slope), float(resample_padding), ptr(u), ptr(None if jitter is None else jitter.reshape(-1)), scale,
         float(near), float(far), ptr(sdist), ptr(tdist), ptr(idx, torch.int32))
    return (sdist, tdist, idx) if want_index else (sdist, tdist)


def encode_ipe(tdist, rays_o, rays_d, radii, basis, embed, ldx: int = 576, out=None):
    B, S1 = tdist.shape
    S = S1 - 1
    X = torch.empty(B * S, ldx, device=tdist.device) if out is None else out
    call("hos_encode_ipe", ptr(tdist), ptr(rays_o), ptr(rays_d), ptr(radii.reshape(-1)), ptr(basis), ptr(embed),
         B, S, ptr(X), ldx)
    return X


def encode_ipe_planes(tdist, rays_o, rays_d, radii, basis, embed, ldx: int = 576, want_bf16: bool = True, want_fp16: bool = True):
    """Same encoder, output directly as Planes (fp16 if want_fp16, bf16 if want_bf16) -- no fp32 copy of the encoding."""
    B, S1 = tdist.shape
    S = S1 - 1
    p16 = Planes.empty(B * S, ldx, torch.float16, tdist.device) if want_fp16 else None
    pb = Planes.empty(B * S, ldx, torch.bfloat16, tdist.device) if want_bf16 else None
    call("hos_encode_ipe_planes", ptr(tdist), ptr(rays_o), ptr(rays_d), ptr(radii.reshape(-1)), ptr(basis), ptr(embed),
         B, S, _pp(p16), _pp(pb), ldx)
    return p16, pb


def encode_viewdirs(viewdirs, S: int, Xv: torch.Tensor, col0: int):
    B = viewdirs.shape[0]
    call("hos_encode_viewdirs", ptr(viewdirs), B, S, ptr(Xv), Xv.stride(0), col0)
    return Xv


class _AlphaWeights(torch.autograd.Function):
    @staticmethod
    def forward(ctx, density, tdist, dirs, opaque):
        B, S = density.shape
        w = torch.empty_like(density)
        call("hos_alpha_weights_fwd", ptr(density), ptr(tdist), ptr(dirs), B, S, int(opaque), ptr(w))
        ctx.save_for_backward(density, tdist, dirs)
        ctx.opaque = int(opaque)
        return w

    @staticmethod
    def backward(ctx, gw):
        density, tdist, dirs = ctx.saved_tensors
        B, S = density.shape
        gd = torch.empty_like(density)
        gw = gw.contiguous()           # bound to a local: a temporary would be freed (and its block reused) before the launch
        call("hos_alpha_weights_bwd", ptr(gw), ptr(density), ptr(tdist), ptr(dirs), B, S, ctx.opaque, ptr(gd))
        return gd, None, None, None


def alpha_weights(density, tdist, dirs, opaque_background: bool):
    """compute_alpha_weights(...)[0] (H:235-261); differentiable w.r.t. density."""
    return _AlphaWeights.apply(density.contiguous(), tdist.contiguous(), dirs.contiguous(), opaque_background)


class _VolRender(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgbs, weights, bg):
        B, S = weights.shape
        out = torch.empty(B, 3, device=weights.device)
        call("hos_volrender_fwd", ptr(rgbs), ptr(weights), B, S, float(bg), ptr(out))
        ctx.save_for_backward(rgbs, weights)
        ctx.bg = float(bg)
        return out

    @staticmethod
    def backward(ctx, g):
        rgbs, weights = ctx.saved_tensors
        B, S = weights.shape
        g_rgbs = torch.empty_like(rgbs) if ctx.needs_input_grad[0] else None
        g_w = torch.empty_like(weights) if ctx.needs_input_grad[1] else None
        g = g.contiguous()
        call("hos_volrender_bwd", ptr(g), ptr(rgbs), ptr(weights), B, S, ctx.bg, ptr(g_rgbs), ptr(g_w))
        return g_rgbs, g_w, None


def volumetric_rendering(rgbs, weights, bg_rgb: float):
    """H:265-275: rgb[B,3]."""
    return _VolRender.apply(rgbs.contiguous(), weights.contiguous(), bg_rgb)


class _Interlevel(torch.autograd.Function):
    """sum over rays and NeRF bins of lossfun_outer(c, w, cp, wp) (H:136-138); grad w.r.t. wp only."""

    @staticmethod
    def forward(ctx, c, w, cp, wp):
        B, Sc = w.shape
        Sp = wp.shape[1]
        per_ray = torch.empty(B, device=w.device)
        call("hos_interlevel_fwd", ptr(c), ptr(w), ptr(cp), ptr(wp), B, Sc, Sp, ptr(per_ray), 0, 0)
        ctx.save_for_backward(c, w, cp, wp)
        return per_ray

    @staticmethod
    def backward(ctx, g):
        c, w, cp, wp = ctx.saved_tensors
        B, Sc = w.shape
        Sp = wp.shape[1]
        gwp = torch.empty_like(wp)
        call("hos_interlevel_bwd", ptr(c), ptr(w), ptr(cp), ptr(wp), B, Sc, Sp, 1.0, ptr(gwp))
        return None, None, None, gwp * g[:, None]


def interlevel_loss_per_ray(c, w, cp, wp):
    return _Interlevel.apply(c.detach().contiguous(), w.detach().contiguous(), cp.detach().contiguous(), wp.contiguous())


def interlevel_indices(c, w, cp, wp):
    """(idx_lo, idx_hi) int32 [B,Sc+1] -- the bit-exact searchsorted indices of H:109-114."""
    B, Sc = w.shape
    Sp = wp.shape[1]
    lo = torch.empty(B, Sc + 1, dtype=torch.int32, device=w.device)
    hi = torch.empty_like(lo)
    per_ray = torch.empty(B, device=w.device)
    call("hos_interlevel_fwd", ptr(c), ptr(w), ptr(cp), ptr(wp), B, Sc, Sp, ptr(per_ray), ptr(lo, torch.int32), ptr(hi, torch.int32))
    return lo, hi


class _Distortion(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, w):
        B, S = w.shape
        per_ray = torch.empty(B, device=w.device)
        call("hos_distortion_fwd", ptr(t), ptr(w), B, S, ptr(per_ray))
        ctx.save_for_backward(t, w)
        return per_ray

    @staticmethod
    def backward(ctx, g):
        t, w = ctx.saved_tensors
        B, S = w.shape
        gw = torch.empty_like(w)
        call("hos_distortion_bwd", ptr(t), ptr(w), B, S, 1.0, ptr(gw))
        return None, gw * g[:, None]


def distortion_loss_per_ray(t, w):
    """H:142-149 per ray; differentiable w.r.t. w (t = sdist is detached in the reference, M:493-494)."""
    return _Distortion.apply(t.detach().contiguous(), w.contiguous())


class _Stage1LossTail(torch.autograd.Function):
    """M1:491-514 behind the per-ray terms: one forward and one backward launch (hos_stage1_loss_{fwd,bwd})."""

    @staticmethod
    def forward(ctx, rgb, target, dist, Sc, mults, inter0, inter1):
        rgb, target, dist = rgb.contiguous(), target.contiguous(), dist.contiguous()
        i0 = None if inter0 is None else inter0.contiguous()
        i1 = None if inter1 is None else inter1.contiguous()
        out = torch.empty(4, device=rgb.device)
        call("hos_stage1_loss_fwd", ptr(rgb), ptr(target), rgb.shape[0], ptr(i0), ptr(i1), Sc, ptr(dist), *mults, ptr(out))
        ctx.save_for_backward(rgb, target, out)
        ctx.Sc, ctx.mults, ctx.have = Sc, mults, (inter0 is not None, inter1 is not None)
        ctx.mark_non_differentiable(out)
        return out[0], out

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        rgb, target, out = ctx.saved_tensors
        B = rgb.shape[0]
        g_rgb = torch.empty_like(rgb)
        g_inter = torch.empty(B, device=rgb.device)
        g_dist = torch.empty(B, device=rgb.device)
        call("hos_stage1_loss_bwd", ptr(rgb), ptr(target), B, ctx.Sc, ptr(out), ptr(g_total.reshape(1).contiguous()), *ctx.mults,
             ptr(g_rgb), ptr(g_inter), ptr(g_dist))
        return g_rgb, None, g_dist, None, None, (g_inter if ctx.have[0] else None), (g_inter if ctx.have[1] else None)


def stage1_loss_tail(rgb, target, inter, dist, Sc, m_data, m_inter, m_dist, pad):
    """(total, [total, mse, interlevel, distortion]) from the per-ray terms `inter` (list of at most two [B] tensors) and `dist` [B]."""
    if len(inter) > 2:
        raise _lib.HosLibraryError("stage1_loss_tail: at most two proposal levels")
    i0 = inter[0] if len(inter) > 0 else None
    i1 = inter[1] if len(inter) > 1 else None
    return _Stage1LossTail.apply(rgb, target, dist, int(Sc), (float(m_data), float(m_inter), float(m_dist), float(pad)), i0, i1)


def head_grad_padded(g_density, density, g_rgb, rgb, rgb_padding, dz_density, col_dd, dz_rgb):
    """head_grad that also zeroes the padding columns of its two operand rows (which may be uninitialised storage)."""
    P = density.numel()
    call("hos_head_grad_padded", ptr(g_density), ptr(density), ptr(g_rgb), ptr(rgb), P, float(rgb_padding),
         ptr(dz_density), 0 if dz_density is None else dz_density.stride(0), col_dd,
         ptr(dz_rgb), 0 if dz_rgb is None else dz_rgb.stride(0))


def state_embed_grad(db, W, c0, N, gb, g_embed):
    """gb[:N] += db[:N] (gb None: skip); g_embed += db[:N] @ W[:N, c0:c0+len(g_embed)] -- one launch (was add_ + mm + add_)."""
    call("hos_state_embed_grad", ptr(db), ptr(W), W.stride(0), c0, N, g_embed.numel(), ptr(gb), ptr(g_embed))


def head_grad(g_density, density, g_rgb, rgb, rgb_padding, dz_density, col_dd, dz_rgb):
    P = density.numel()
    call("hos_head_grad", ptr(g_density), ptr(density), ptr(g_rgb), ptr(rgb), P, float(rgb_padding),
         ptr(dz_density), 0 if dz_density is None else dz_density.stride(0), col_dd,
         ptr(dz_rgb), 0 if dz_rgb is None else dz_rgb.stride(0))


# ------------------------------------------------------------------------------------------ optimiser
def sumsq(g: torch.Tensor, out: torch.Tensor):
    call("hos_sumsq", ptr(g), g.numel(), ptr(out))


def sumsq_blocks() -> int:
    return int(_lib.load().hos_sumsq_blocks())


def sumsq_partials(spans, partial: torch.Tensor):
    """partial[:sumsq_blocks()] = per-block sums of squares over all `spans` (<= 8 fp32 tensors, numel % 4 == 0), one launch."""
    import ctypes
    call("hos_sumsq_partials", len(spans), _table([ptr(t) for t in spans]), _table([t.numel() for t in spans], ctypes.c_longlong), ptr(partial))


def adam_multi(spans, step: int, beta1: float, beta2: float, eps: float, grad_scale: float, partial, max_norm: float, guard=(None, None)):
    """torch.optim.Adam over up to 32 spans in one launch.  spans: (p, g, m, v, hyper | None, lr[, lazy | None[, row length]]) -- hyper: device [3] row
    {lr, 1-b1^t, 1/sqrt(1-b2^t)} (graph replay), else lr / `step` from the host; lazy: the span's [4] state row of `adam_lazy_prepare`
    (an identically-zero gradient = torch's `grad is None`: the span is skipped and counts its own steps).
    guard: (range word, skipped counter) or Nones."""
    import ctypes
    n = len(spans)
    col = lambda i: _table([ptr(sp[i]) for sp in spans])
    lazy = _table([ptr(sp[6]) if len(sp) > 6 else None for sp in spans]) if any(len(sp) > 6 and sp[6] is not None for sp in spans) else None
    rows = _table([int(sp[7]) if len(sp) > 7 and sp[7] else 0 for sp in spans], ctypes.c_int) if lazy is not None else None
    call("hos_adam_multi_lazy", n, col(0), col(1), col(2), col(3), _table([sp[0].numel() for sp in spans], ctypes.c_longlong),
         col(4), lazy, rows, _table([float(sp[5]) for sp in spans], ctypes.c_float), int(step), float(beta1), float(beta2), float(eps),
         float(grad_scale), ptr(partial), float(max_norm), ptr(guard[0], torch.int32), ptr(guard[1], torch.int32))


def adam_lazy_prepare(grads, states, beta1: float, beta2: float, guard=None):
    """One launch: for every lazily updated span (grads[i]: its slice of the reduced flat gradient, states[i]: its [4] fp32 state row)
    decide active / inactive (gradient identically zero) and advance the span's own step count (hos_adam_lazy_prepare)."""
    import ctypes
    for i in range(0, len(grads), 32):                     # (<= 32 spans per launch)
        g_, s_ = grads[i:i + 32], states[i:i + 32]
        call("hos_adam_lazy_prepare", len(g_), _table([ptr(g) for g in g_]), _table([g.numel() for g in g_], ctypes.c_longlong),
             _table([ptr(s) for s in s_]), float(beta1), float(beta2), ptr(guard, torch.int32))


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0, sumsq_buf=None, max_norm=0.0):
    call("hos_adam_step", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(lr), float(beta1), float(beta2),
         float(eps), int(step), float(grad_scale), ptr(sumsq_buf), float(max_norm))


# ------------------------------------------------------------------------------------------ human branch
_T_CACHE = {}


def _t_vals(N: int, device) -> torch.Tensor:
    key = (N, str(device))
    if key not in _T_CACHE:
        _T_CACHE[key] = torch.linspace(0.0, 1.0, steps=N).to(device)      # N:410, built by torch for bit-exactness
    return _T_CACHE[key]


def human_sample_warp(rays_o, rays_d, near, far, N: int, R, T, vol, bbox_min, bbox_scale, t_rand=None, K: int = 26):
    """(z_vals [B,N], pts [B,N,3], x_skel [B*N,3], mask [B*N]) -- N:409-424, N:451, N:304-355 fused."""
    B = rays_o.shape[0]
    dev = rays_o.device
    z = torch.empty(B, N, device=dev)
    pts = torch.empty(B, N, 3, device=dev)
    x_skel = torch.empty(B * N, 3, device=dev)
    mask = torch.empty(B * N, device=dev)
    V = vol.shape[-1]
    call("hos_human_sample_warp", ptr(rays_o), ptr(rays_d), ptr(near.reshape(-1)), ptr(far.reshape(-1)),
         ptr(_t_vals(N, dev)), ptr(None if t_rand is None else t_rand.reshape(-1)), ptr(R), ptr(T), ptr(vol), V,
         ptr(bbox_min), ptr(bbox_scale), B, N, K, ptr(z), ptr(pts), ptr(x_skel), ptr(mask))
    return z, pts, x_skel, mask


def lbs_forward(cnl_pts, R_f, T_f, vol_cl, bbox_min, bbox_scale, K: int = 26, rows_dev=None):
    P = cnl_pts.shape[0]
    out = torch.empty(P, 3, device=cnl_pts.device)
    V, CL = vol_cl.shape[0], vol_cl.shape[-1]
    call("hos_lbs_forward", ptr(cnl_pts), ptr(R_f), ptr(T_f), ptr(vol_cl), V, CL, ptr(bbox_min), ptr(bbox_scale), P, K, ptr(out),
         ptr(rows_dev, torch.int32))
    return out


def embed_hannw(x, band_w, cond, E, PE=None, rows_dev=None):
    P = x.shape[0]
    call("hos_embed_hannw", ptr(x), ptr(band_w), band_w.numel(), ptr(cond), 0 if cond is None else cond.numel(), P,
         ptr(E), E.stride(0), ptr(PE), 0 if PE is None else PE.stride(0), ptr(rows_dev, torch.int32))


def embed_fourier(x, num_freqs, state, E, E2=None):
    P = x.shape[0]
    call("hos_embed_fourier", ptr(x), num_freqs, ptr(state), 0 if state is None else state.numel(), P,
         ptr(E), E.stride(0), ptr(E2), 0 if E2 is None else E2.stride(0))


# ------------------------------------------------------------------------------------------ composites
class _Raw2Outputs(torch.autograd.Function):
    """rgb [B,3] (+ weights [B,S]) from packed rgb-sigma [B,S,4]; differentiable w.r.t. rgbsigma and mask."""

    @staticmethod
    def forward(ctx, rgbsigma, z_vals, rays_d, mask, bgcolor, last_dist):
        B, S = z_vals.shape
        dev = z_vals.device
        rgb = torch.empty(B, 3, device=dev)
        acc = torch.empty(B, device=dev)
        w = torch.empty(B, S, device=dev)
        depth = torch.empty(B, device=dev)
        call("hos_raw2outputs_fwd", ptr(rgbsigma), 4, ptr(rgbsigma) + 12, 4, ptr(z_vals), ptr(rays_d), ptr(mask), ptr(bgcolor),
             float(last_dist), B, S, ptr(rgb), ptr(acc), ptr(w), ptr(depth))
        ctx.save_for_backward(rgbsigma, z_vals, rays_d, mask, bgcolor)
        ctx.last_dist = float(last_dist)
        ctx.mark_non_differentiable(acc, depth)
        ctx.set_materialize_grads(False)       # no zero tensors for the cotangents nobody feeds (acc, depth: one fill launch each)
        return rgb, acc, w, depth

    @staticmethod
    def backward(ctx, g_rgb, g_acc, g_w, g_depth):
        rgbsigma, z_vals, rays_d, mask, bgcolor = ctx.saved_tensors
        B, S = z_vals.shape
        g_rs = torch.empty_like(rgbsigma)
        g_mask = torch.empty(B, S, device=z_vals.device) if mask is not None else None
        g_rgb = zeros((rgbsigma.shape[0], 3), rgbsigma.device) if g_rgb is None else g_rgb.contiguous()     # locals keep the cotangents alive until the launch is enqueued
        g_w = None if g_w is None else g_w.contiguous()
        call("hos_raw2outputs_bwd", ptr(g_rgb), ptr(g_w), ptr(rgbsigma), 4,
             ptr(rgbsigma) + 12, 4, ptr(z_vals), ptr(rays_d), ptr(mask), ptr(bgcolor), ctx.last_dist, B, S,
             ptr(g_rs), 4, ptr(g_rs) + 12, 4, ptr(g_mask))
        return g_rs, None, None, g_mask, None, None


def raw2outputs(rgbsigma, z_vals, rays_d, mask=None, bgcolor=None, last_dist: float = 1e10):
    """M:73-99 on activated samples.  Returns (rgb_map, acc_map, weights, depth_map)."""
    return _Raw2Outputs.apply(rgbsigma.contiguous(), z_vals.contiguous(), rays_d.contiguous(),
                              None if mask is None else mask.contiguous(),
                              None if bgcolor is None else bgcolor.contiguous().float(), last_dist)


class _MergeComposite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bkg_rgb, bkg_density, human_rgbsigma, pts_mask, bkg_tdist, pts, rays_o, rays_d, A, tiny_flag, thre_fg):
        B, Sb = bkg_density.shape
        Sh = pts_mask.shape[1]
        dev = bkg_density.device
        rgb = torch.empty(B, 3, device=dev)
        idx_fg = torch.empty(B, dtype=torch.int32, device=dev)
        order = torch.empty(B, Sb + Sh, dtype=torch.int32, device=dev)
        hw = torch.empty(B, Sh, device=dev)
        zh = torch.empty(B, Sh, device=dev)
        call("hos_merge_composite_fwd", ptr(bkg_tdist), ptr(bkg_rgb), ptr(bkg_density), ptr(human_rgbsigma), ptr(pts),
             ptr(pts_mask), ptr(rays_o), ptr(rays_d), ptr(A), ptr(tiny_flag, torch.int32), B, Sb, Sh, float(thre_fg),
             ptr(rgb), ptr(idx_fg, torch.int32), ptr(order, torch.int32), ptr(hw), ptr(zh))
        ctx.save_for_backward(bkg_rgb, bkg_density, human_rgbsigma, pts_mask, bkg_tdist, pts, rays_o, rays_d, A, tiny_flag)
        ctx.thre = float(thre_fg)
        ctx.mark_non_differentiable(idx_fg, order, zh)
        ctx.set_materialize_grads(False)       # idx_fg / order / zh carry no gradient: no zero tensors for them (3 fills per step)
        return rgb, hw, idx_fg, order, zh

    @staticmethod
    def backward(ctx, g_rgb, g_hw, *_):
        bkg_rgb, bkg_density, human, mask, tdist, pts, ro, rd, A, flag = ctx.saved_tensors
        B, Sb = bkg_density.shape
        Sh = mask.shape[1]
        g_brgb = torch.empty_like(bkg_rgb)
        g_bden = torch.empty_like(bkg_density)
        g_h = torch.empty_like(human)
        g_m = torch.empty_like(mask)
        g_rgb = zeros((B, 3), mask.device) if g_rgb is None else g_rgb.contiguous()
        g_hw = None if g_hw is None else g_hw.contiguous()
        call("hos_merge_composite_bwd", ptr(g_rgb), ptr(g_hw),
             ptr(tdist), ptr(bkg_rgb), ptr(bkg_density), ptr(human), ptr(pts), ptr(mask), ptr(ro), ptr(rd), ptr(A),
             ptr(flag, torch.int32), B, Sb, Sh, ctx.thre, ptr(g_brgb), ptr(g_bden), ptr(g_h), ptr(g_m))
        return g_brgb, g_bden, g_h, g_m, None, None, None, None, None, None, None


def merge_composite(bkg_tdist, bkg_rgb, bkg_density, human_rgbsigma, newsmpl_pts, pts_mask, rays_o_bkg, rays_d_bkg,
                    newsmpl_to_scale_world, thre_fg: float = 5e-3):
    """Stage-3 inline composite (M:1524-1596).  Returns (rgb [B,3], human_weights_sorted [B,Sh], idx_fg [B] int32,
    total_order [B,Sb+Sh] int32, z_human [B,Sh]).  No host synchronisation: the `any |d| < 1e-5` test of M:1526
    stays on the device."""
    rd = rays_d_bkg.contiguous()
    tiny = torch.empty(1, dtype=torch.int32, device=rd.device)
    call("hos_any_abs_below", ptr(rd), rd.numel(), 1e-5, ptr(tiny, torch.int32))
    return _MergeComposite.apply(bkg_rgb.contiguous(), bkg_density.contiguous(), human_rgbsigma.contiguous(),
                                 pts_mask.contiguous(), bkg_tdist.contiguous(), newsmpl_pts.contiguous(),
                                 rays_o_bkg.contiguous(), rd, newsmpl_to_scale_world.contiguous().float(), tiny, thre_fg)


# ------------------------------------------------------------------------------------------ per-frame prologue (P2, P3)
POSE_PARAM_ORDER = ("block_mlps.0", "block_mlps.2", "block_mlps.4", "block_mlps_dstR.0", "block_mlps_dstR.2", "block_mlps_dstT.0",
                    "block_mlps_dstT.2")          # (weight, bias) pairs in this order = the 14 pointers of hos_pose_refine_*


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[ptr(t) for t in tensors])


class _PoseRefine(torch.autograd.Function):
    """(Rs', Ts') = refine(Rs, Ts, posevec) for F frames: BodyPoseRefiner + Rodrigues + N:589-605 in one launch; the
    backward accumulates the pose decoder's parameter gradients straight into `grads` (views of the flat gradient buffer)."""

    @staticmethod
    def forward(ctx, token, Rs, Ts, posevec, weights, grads):
        F_, K = Rs.shape[0], Rs.shape[1]
        Rs, Ts, posevec = Rs.contiguous(), Ts.contiguous(), posevec.contiguous()
        Ro, To = torch.empty_like(Rs), torch.empty_like(Ts)
        saved = torch.empty(F_, int(_lib.load().hos_pose_refine_saved_floats()), device=Rs.device)
        width = weights[0].shape[0]
        call("hos_pose_refine_fwd", ptr(posevec), ptr(Rs), ptr(Ts), _ptr_array(weights), F_, K, width, ptr(Ro), ptr(To), ptr(saved))
        ctx.save_for_backward(Rs, posevec, saved)
        ctx.weights, ctx.grads, ctx.dims = weights, grads, (F_, K, width)
        return Ro, To

    @staticmethod
    def backward(ctx, gR, gT):
        Rs, posevec, saved = ctx.saved_tensors
        F_, K, width = ctx.dims
        gR = zeros(tuple(Rs.shape), Rs.device) if gR is None else gR.contiguous()
        gT = zeros((F_, K, 3), Rs.device) if gT is None else gT.contiguous()
        ws = torch.empty(F_, int(_lib.load().hos_pose_refine_workspace_floats()), device=Rs.device)
        call("hos_pose_refine_bwd", ptr(gR), ptr(gT), ptr(posevec), ptr(Rs), ptr(saved), _ptr_array(ctx.weights), _ptr_array(ctx.grads),
             F_, K, width, ptr(ws))
        return None, None, None, None, None, None


def pose_refine(token, Rs, Ts, posevec, weights, grads):
    """weights / grads: the 14 pose-decoder tensors (POSE_PARAM_ORDER x (weight, bias)) and their gradient buffers."""
    return _PoseRefine.apply(token, Rs, Ts, posevec, list(weights), list(grads))


class _MotionBasis(torch.autograd.Function):
    """U:134-174 for F frames in one launch: (R_bwd, T_bwd, R_fwd, T_fwd), each [F,K,...]; differentiable w.r.t. Rs, Ts."""

    @staticmethod
    def forward(ctx, Rs, Ts, cnl_gtfms):
        F_, K = Rs.shape[0], Rs.shape[1]
        Rs, Ts, cnl = Rs.contiguous(), Ts.contiguous(), cnl_gtfms.contiguous()
        dev = Rs.device
        Rb, Tb = torch.empty(F_, K, 3, 3, device=dev), torch.empty(F_, K, 3, device=dev)
        Rf, Tf = torch.empty(F_, K, 3, 3, device=dev), torch.empty(F_, K, 3, device=dev)
        call("hos_motion_basis_fwd", ptr(Rs), ptr(Ts), ptr(cnl), F_, K, ptr(Rb), ptr(Tb), ptr(Rf), ptr(Tf))
        ctx.save_for_backward(Rs, Ts, cnl)
        return Rb, Tb, Rf, Tf

    @staticmethod
    def backward(ctx, gRb, gTb, gRf, gTf):
        Rs, Ts, cnl = ctx.saved_tensors
        F_, K = Rs.shape[0], Rs.shape[1]
        # locals, not temporaries: four materialised cotangents freed one after the other would share one allocator block
        gRb, gTb, gRf, gTf = (None if g is None else g.contiguous() for g in (gRb, gTb, gRf, gTf))
        gRs, gTs = torch.empty_like(Rs), torch.empty_like(Ts)
        call("hos_motion_basis_bwd", ptr(gRb), ptr(gTb), ptr(gRf), ptr(gTf), ptr(Rs), ptr(Ts), ptr(cnl), F_, K, ptr(gRs), ptr(gTs))
        return gRs, gTs, None


def motion_basis(Rs, Ts, cnl_gtfms):
    return _MotionBasis.apply(Rs, Ts, cnl_gtfms)


# ------------------------------------------------------------------------------------------ cycle-set selection (P9)
_COMPACT_WS = {}


def _compact_workspace(device) -> torch.Tensor:
    key = _stream_key(device)
    if key not in _COMPACT_WS:
        _COMPACT_WS[key] = torch.zeros(int(_lib.load().hos_compact_workspace_ints()), dtype=torch.int32, device=device)
    return _COMPACT_WS[key]


class _CompactRows(torch.autograd.Function):
    """Order-preserving selection of the rows with mask > thr into fixed-capacity buffers (N:505-536 without the boolean
    index): returns (a_sel [P,3], b_sel [P,3], sel [P] int32, count [1] int32); rows >= count are zero.  Differentiable
    w.r.t. `a` (the canonical points): the gradient is scattered back to the selected rows."""

    @staticmethod
    def forward(ctx, mask, thr, a, b):
        P = mask.numel()
        dev = mask.device
        a_sel, b_sel = torch.empty(P, 3, device=dev), torch.empty(P, 3, device=dev)
        sel = torch.empty(P, dtype=torch.int32, device=dev)
        count = torch.empty(1, dtype=torch.int32, device=dev)
        call("hos_compact_rows", ptr(mask), float(thr), ptr(a), ptr(b), P, ptr(count, torch.int32), ptr(sel, torch.int32), ptr(a_sel),
             ptr(b_sel), ptr(_compact_workspace(dev), torch.int32))
        ctx.save_for_backward(sel, count)
        ctx.mark_non_differentiable(b_sel, sel, count)
        ctx.set_materialize_grads(False)       # b_sel [P,3] / sel / count carry no gradient: no zero tensors for them
        return a_sel, b_sel, sel, count

    @staticmethod
    def backward(ctx, g_a, *_):
        if g_a is None:
            return None, None, None, None
        sel, count = ctx.saved_tensors
        P = sel.numel()
        g = torch.empty(P, 3, device=sel.device)
        g_a = g_a.contiguous()
        call("hos_scatter_rows", ptr(g_a), ptr(sel, torch.int32), ptr(count, torch.int32), P, ptr(g))
        return None, None, g, None


def compact_rows(mask, thr: float, a, b):
    return _CompactRows.apply(mask.detach().reshape(-1).contiguous(), thr, a.reshape(-1, 3).contiguous(), b.detach().reshape(-1, 3).contiguous())


# ------------------------------------------------------------------------------------------ training losses (C4)
_LOSS_WS = {}


def _loss_workspace(device) -> torch.Tensor:
    """Block partials + completion ticket of hos_train_losses_fwd (zero-initialised once: the kernel re-arms the ticket)."""
    key = _stream_key(device)
    if key not in _LOSS_WS:
        _LOSS_WS[key] = torch.zeros(int(_lib.load().hos_train_losses_workspace_floats()), device=device)
    return _LOSS_WS[key]


class _TrainLosses(torch.autograd.Function):
    """total = w_mse*mse + w_flow*flow + w_cycle*cycle (M:1690-1716 / M2:918-944) in one launch, gradients in another.
    Inputs that are None switch their term off.  Returns (total [], parts [8] = total, mse, flow, cycle, ...)."""

    @staticmethod
    def forward(ctx, rgb, target, mse_const, mse_count, pts_prev, weights, ray_grid, fg, cam, Kin, observe, deform, n_cyc_dev,
                w_mse, w_flow, w_cycle):
        B = rgb.shape[0]
        S = 0 if pts_prev is None else pts_prev.shape[1]
        n_cyc = 0 if observe is None else observe.shape[0]
        out = torch.empty(8, device=rgb.device)
        call("hos_train_losses_fwd", ptr(rgb), ptr(target), B, float(mse_const), float(mse_count), ptr(pts_prev), ptr(weights),
             ptr(ray_grid), ptr(fg, torch.int32), ptr(cam), ptr(Kin), S, ptr(observe), ptr(deform), n_cyc, ptr(n_cyc_dev, torch.int32),
             float(w_mse), float(w_flow), float(w_cycle), ptr(_loss_workspace(rgb.device)), ptr(out))
        ctx.save_for_backward(rgb, target, pts_prev, weights, ray_grid, fg, cam, Kin, observe, deform, n_cyc_dev, out)
        ctx.cfg = (float(mse_count), float(w_mse), float(w_flow), float(w_cycle))
        parts = out.detach().clone()
        ctx.mark_non_differentiable(parts)
        ctx.set_materialize_grads(False)
        return out[0], parts

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        if g_total is None:
            return (None,) * 16
        rgb, target, pts_prev, weights, ray_grid, fg, cam, Kin, observe, deform, n_cyc_dev, out = ctx.saved_tensors
        mse_count, w_mse, w_flow, w_cycle = ctx.cfg
        B = rgb.shape[0]
        S = 0 if pts_prev is None else pts_prev.shape[1]
        n_cyc = 0 if observe is None else observe.shape[0]
        need = ctx.needs_input_grad
        g_rgb = torch.empty_like(rgb) if need[0] else None
        g_pts = torch.empty_like(pts_prev) if (pts_prev is not None and need[4]) else None
        g_w = torch.empty_like(weights) if (weights is not None and need[5]) else None
        g_def = torch.empty_like(deform) if (deform is not None and need[11]) else None
        g_total = g_total.contiguous()
        call("hos_train_losses_bwd", ptr(g_total), ptr(out), ptr(rgb), ptr(target), B, mse_count, ptr(pts_prev), ptr(weights),
             ptr(ray_grid), ptr(fg, torch.int32), ptr(cam), ptr(Kin), S, ptr(observe), ptr(deform), n_cyc, ptr(n_cyc_dev, torch.int32),
             w_mse, w_flow, w_cycle, ptr(g_rgb), ptr(g_pts), ptr(g_w), ptr(g_def))
        return (g_rgb, None, None, None, g_pts, g_w, None, None, None, None, None, g_def, None, None, None, None)


def train_losses(rgb, target, mse_const=0.0, mse_count=None, pts_prev=None, weights=None, ray_grid=None, fg=None, cam_prev=None,
                 intrinsics_prev=None, observe=None, deform=None, n_cyc_dev=None, w_mse=0.2, w_flow=0.01, w_cycle=0.01):
    """Returns (total, parts[8]) with parts = [total, mse, flow, cycle (unweighted), 1/flow-denominator, 1/n_cyc, sum M, n_cyc]."""
    c = lambda t: None if t is None else t.contiguous()
    mse_count = float(rgb.numel()) if mse_count is None else mse_count
    if pts_prev is None:
        weights = ray_grid = fg = cam_prev = intrinsics_prev = None
    return _TrainLosses.apply(c(rgb), c(target).to(rgb.dtype), mse_const, mse_count, c(pts_prev), c(weights), c(ray_grid),
                              None if fg is None else fg.to(torch.int32).contiguous(), c(cam_prev), c(intrinsics_prev),
                              c(observe), c(deform), n_cyc_dev, w_mse, w_flow, w_cycle)


# ------------------------------------------------------------------------------------------ human branch, backward
def _zeros_like_many(*ts):
    """Zeroed accumulators shaped like `ts` out of ONE allocation and ONE fill launch (every launch of a replayed step costs
    4-5 us whatever its size; sizes are rounded to 16 bytes so that every view stays aligned)."""
    return zeros_many([tuple(t.shape) for t in ts], ts[0].device, ts[0].dtype)


class _UnbindFrames(torch.autograd.Function):
    """x [F, ...] -> (x[0], ..., x[F-1]).  The backward is one stack instead of a zero fill + slice copy per frame and an add per
    extra frame (autograd's select backward); a single frame costs no launch at all."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return tuple(x[f] for f in range(x.shape[0]))

    @staticmethod
    def backward(ctx, *gs):
        if all(g is None for g in gs):
            return None
        ref = next(g for g in gs if g is not None)
        if len(gs) == 1:
            return gs[0].unsqueeze(0)
        return torch.stack([torch.zeros_like(ref) if g is None else g for g in gs], 0)


def unbind_frames(x: torch.Tensor):
    return _UnbindFrames.apply(x)


class _UnbindFramesMany(torch.autograd.Function):
    """(x_0 [F, ...], ..., x_{m-1} [F, ...]) -> the F per-frame views of each, frame-major per tensor.  Backward: ONE launch stacks
    (or zero-fills) all m x F cotangents into one allocation (per tensor that was a torch.stack + a zero fill per missing frame)."""

    @staticmethod
    def forward(ctx, *xs):
        ctx.shapes = [tuple(x.shape) for x in xs]
        ctx.set_materialize_grads(False)
        out = []
        for x in xs:
            out += [x[f] for f in range(x.shape[0])]
        return tuple(out)

    @staticmethod
    def backward(ctx, *gs):
        if all(g is None for g in gs):
            return (None,) * len(ctx.shapes)
        dev = next(g for g in gs if g is not None).device
        outs = [torch.empty(sh, device=dev) for sh in ctx.shapes] if len(ctx.shapes) > 8 else None
        sizes = [int(np.prod(sh)) for sh in ctx.shapes]
        buf = torch.empty(sum((n + 3) // 4 * 4 for n in sizes), device=dev)
        outs, o = [], 0
        for sh, n in zip(ctx.shapes, sizes):
            outs.append(buf[o:o + n].view(sh))
            o += (n + 3) // 4 * 4
        dsts, srcs, k = [], [], 0
        for x_out in outs:
            for f in range(x_out.shape[0]):
                dsts.append(x_out[f])
                g = gs[k]
                srcs.append(None if g is None else g.contiguous())
                k += 1
        copy_or_zero_n(dsts, srcs)
        return tuple(outs)


def unbind_frames_many(*xs):
    """Per-frame views of several [F, ...] tensors: returns a list of tuples, one per tensor."""
    F_ = xs[0].shape[0]
    if F_ == 1 or not (torch.is_grad_enabled() and any(x.requires_grad for x in xs)):
        return [tuple(x[f] for f in range(x.shape[0])) for x in xs]
    flat = _UnbindFramesMany.apply(*xs)
    out, k = [], 0
    for x in xs:
        out.append(tuple(flat[k:k + x.shape[0]]))
        k += x.shape[0]
    return out


SAMPLE_WARP_BWD_REUSE = os.environ.get("HOS_SAMPLE_WARP_BWD_REUSE", "1") == "1"   # A/B switch: backward reads the forward's x_skel / mask


class _SampleWarp(torch.autograd.Function):
    """(z, pts, x_skel, mask) with gradients to the motion-weight volume and the backward motion basis."""

    @staticmethod
    def forward(ctx, vol, R, T, rays_o, rays_d, near, far, N, bmin, bscale, t_rand, K):
        z, pts, x_skel, mask = human_sample_warp(rays_o, rays_d, near, far, N, R, T, vol, bmin, bscale, t_rand, K)
        # x_skel / mask are kept for the backward kernel (it would otherwise re-evaluate the 26 x 8 taps per point they came from)
        ctx.save_for_backward(vol, R, T, pts, bmin, bscale, x_skel, mask)
        ctx.K = K
        ctx.mark_non_differentiable(z, pts)
        ctx.set_materialize_grads(False)       # z [B,N] / pts [B,N,3] carry no gradient: no zero tensors for them (2 fills per step)
        return z, pts, x_skel, mask

    @staticmethod
    def backward(ctx, gz, gpts, g_xskel, g_mask):
        vol, R, T, pts, bmin, bscale, x_skel, mask = ctx.saved_tensors
        K = ctx.K
        P = pts.shape[0] * pts.shape[1]
        g_vol, g_R, g_T = _zeros_like_many(vol, R, T)
        gx = zeros((P, 3), pts.device) if g_xskel is None else g_xskel.contiguous()
        gm = zeros(P, pts.device) if g_mask is None else g_mask.contiguous()
        scratch = torch.empty(P, 2, device=pts.device)
        call("hos_human_sample_warp_bwd", ptr(pts), ptr(R), ptr(T), ptr(vol), vol.shape[-1], ptr(bmin), ptr(bscale), P, K,
             ptr(gx), ptr(gm), ptr(g_vol), ptr(g_R), ptr(g_T), ptr(scratch),
             ptr(x_skel) if SAMPLE_WARP_BWD_REUSE else None, ptr(mask) if SAMPLE_WARP_BWD_REUSE else None)
        return g_vol, g_R, g_T, None, None, None, None, None, None, None, None, None


def human_sample_warp_ad(vol, R, T, rays_o, rays_d, near, far, N, bmin, bscale, t_rand=None, K: int = 26):
    return _SampleWarp.apply(vol.contiguous(), R.contiguous(), T.contiguous(), rays_o, rays_d, near, far, N, bmin, bscale, t_rand, K)


class _LbsForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cnl, vol_cl, R_f, T_f, bmin, bscale, K, rows_dev):
        out = lbs_forward(cnl, R_f, T_f, vol_cl, bmin, bscale, K, rows_dev)
        ctx.save_for_backward(cnl, vol_cl, R_f, T_f, bmin, bscale, rows_dev)
        ctx.K = K
        return out

    @staticmethod
    def backward(ctx, g):
        cnl, vol_cl, R_f, T_f, bmin, bscale, rows_dev = ctx.saved_tensors
        P = cnl.shape[0]
        g_cnl = torch.empty_like(cnl)
        g_vol, g_R, g_T = _zeros_like_many(vol_cl, R_f, T_f)
        g = g.contiguous()
        call("hos_lbs_forward_bwd", ptr(cnl), ptr(R_f), ptr(T_f), ptr(vol_cl), vol_cl.shape[0], vol_cl.shape[-1], ptr(bmin),
             ptr(bscale), P, ctx.K, ptr(g), ptr(g_cnl), ptr(g_vol), ptr(g_R), ptr(g_T), ptr(rows_dev, torch.int32))
        return g_cnl, g_vol, g_R, g_T, None, None, None, None


def lbs_forward_ad(cnl, vol_cl, R_f, T_f, bmin, bscale, K: int = 26, rows_dev=None):
    return _LbsForward.apply(cnl.contiguous(), vol_cl.contiguous(), R_f.contiguous(), T_f.contiguous(), bmin, bscale, K, rows_dev)


def embed_bwd(x, band_w, num_freqs, identity, dA, colA, dB, colB, g_x, accumulate, rows_dev=None):
    call("hos_embed_bwd", ptr(x), ptr(band_w), num_freqs, int(identity), ptr(dA), dA.stride(0), colA,
         ptr(dB), 0 if dB is None else dB.stride(0), colB, x.shape[0], ptr(g_x), int(accumulate), ptr(rows_dev, torch.int32))


def embed_bwd_res(x, band_w, num_freqs, identity, dA, colA, dB, colB, res, rows_dev=None):
    """g_x [P,3] = res + d(features)/dx in one launch (rows past *rows_dev: res alone) -- no clone of the residual cotangent."""
    g_x = torch.empty(x.shape[0], 3, device=x.device)
    call("hos_embed_bwd_res", ptr(x), ptr(band_w), num_freqs, int(identity), ptr(dA), dA.stride(0), colA,
         ptr(dB), 0 if dB is None else dB.stride(0), colB, x.shape[0], ptr(res), ptr(g_x), ptr(rows_dev, torch.int32))
    return g_x


def slice_mask(src, col0, mask_src, mcol0, width, out, rows_dev=None):
    call("hos_slice_mask", ptr(src), src.stride(0), col0, ptr(mask_src), 0 if mask_src is None else mask_src.stride(0), mcol0,
         src.shape[0], width, ptr(out), out.stride(0), ptr(rows_dev, torch.int32))


def slice_pad(src, col0, width, out, rows_dev=None):
    """out[:, :] = [src[:, col0:col0+width] | 0 ...] over whole rows of `out` (uninitialised storage is fine)."""
    call("hos_slice_pad", ptr(src), src.stride(0), col0, src.shape[0], width, ptr(out), out.stride(0), ptr(rows_dev, torch.int32))


def rgbsigma_grad(g, y, dz):
    call("hos_rgbsigma_grad", ptr(g), ptr(y), y.shape[0], ptr(dz), dz.stride(0))


def adam_step_dyn(p, g, m, v, hyper, beta1, beta2, eps, grad_scale=1.0, sumsq_buf=None, max_norm=0.0):
    """Adam with (lr, bias corrections) read from the device tensor `hyper` [3] -- graph-replay friendly."""
    call("hos_adam_step_dyn", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(hyper), float(beta1), float(beta2),
         float(eps), float(grad_scale), ptr(sumsq_buf), float(max_norm))


# ------------------------------------------------------------------------------------------ planes GEMMs
class Planes:
    """A matrix [rows, ld] stored as interleaved 16-bit planes: one [rows, ld/32, 2, 32] tensor -- per row and
    32-column block the 32 hi values then the 32 lo values (value = hi + lo).  dtype torch.float16 for forward
    operands, torch.bfloat16 for gradient-side operands.  `ld` is the logical (padded) column count, % 32 == 0.
    `bits` (optional): the ReLU bit mask of the matrix, written by linearp_fwd and read by linearp_dgrad -- int32
    [ceil(rows/32), ceil(ld/64), 64] (include/hosrender.h, hos_linearp_fwd)."""

    __slots__ = ("t", "rows", "cols", "bits")

    def __init__(self, t: torch.Tensor, rows: int, cols: int, bits: Optional[torch.Tensor] = None):
        assert t.dim() == 4 and t.shape[2] == 2 and t.shape[3] == 32
        self.t, self.rows, self.cols, self.bits = t, rows, cols, bits

    @staticmethod
    def empty(rows: int, ld: int, dtype, device, cols: Optional[int] = None, relu_bits: bool = False):
        assert ld % 32 == 0, "planes need a leading dimension that is a multiple of 32"
        bits = torch.empty((rows + 31) // 32, (ld + 63) // 64, 64, dtype=torch.int32, device=device) if relu_bits else None
        return Planes(torch.empty(rows, ld // 32, 2, 32, dtype=dtype, device=device), rows, ld if cols is None else cols, bits)

    @staticmethod
    def from_flat(flat16: torch.Tensor, offset: int, rows: int, ld: int, cols: Optional[int] = None):
        """View of `rows x ld` elements starting at logical element `offset` (% 32 == 0) of a flat planes buffer."""
        return Planes(flat16[2 * offset:2 * (offset + rows * ld)].view(rows, ld // 32, 2, 32), rows, ld if cols is None else cols)

    @property
    def hi(self):
        return self.t[:, :, 0, :].reshape(self.rows, -1)

    @property
    def lo(self):
        return self.t[:, :, 1, :].reshape(self.rows, -1)

    @property
    def ld(self) -> int:
        return self.t.shape[1] * 32

    def float(self) -> torch.Tensor:
        return (self.t[:, :, 0, :].float() + self.t[:, :, 1, :].float()).reshape(self.rows, -1)[:, :self.cols]


def _pp(x):
    """device pointer of a Planes object (contiguous 16-bit HIP tensor) or 0."""
    if x is None:
        return 0
    t = x.t if isinstance(x, Planes) else x
    if not (t.is_cuda and t.is_contiguous() and t.element_size() == 2):
        raise _lib.HosLibraryError("expected a contiguous 16-bit HIP planes tensor")
    return t.data_ptr()


def split_planes(src: torch.Tensor, C: Optional[int] = None, dtype=torch.float16, ldo: Optional[int] = None,
                 transposed: bool = False, ldt: Optional[int] = None, row_major: bool = True):
    """fp32 [R, lds] -> (row-major Planes [R][ldo] | None, transposed Planes [C][ldt] | None)."""
    R = src.shape[0]
    C = src.shape[1] if C is None else C
    dev = src.device
    out = outT = None
    if row_major:
        ldo = round_up(C, 32) if ldo is None else ldo
        out = Planes.empty(R, ldo, dtype, dev, C)
    if transposed:
        ldt = round_up(R, 32) if ldt is None else ldt
        outT = Planes.empty(C, ldt, dtype, dev, R)
    call("hos_split_planes", ptr(src), src.stride(0), R, C, 0 if dtype == torch.float16 else 1,
         _pp(out), 0 if out is None else out.ld, _pp(outT), 0 if outT is None else outT.ld)
    return out, outT


def split_planes_T_batch(mats):
    """Transposed bf16 planes [C][round_up(R, 32)] of every fp32 matrix [R, C(ld)] in `mats`, 12 per launch (one launch per MLP
    call instead of one per layer)."""
    import ctypes
    outs = []
    for i0 in range(0, len(mats), 12):
        chunk = mats[i0:i0 + 12]
        planes = [Planes.empty(m.shape[1], round_up(m.shape[0], 32), torch.bfloat16, m.device, m.shape[0]) for m in chunk]
        n = len(chunk)
        src = (ctypes.c_void_p * n)(*[ptr(m) for m in chunk])
        out = (ctypes.c_void_p * n)(*[_pp(p) for p in planes])
        ia = lambda vals: (ctypes.c_int * n)(*vals)
        call("hos_split_planes_t_batch", n, src, ia([m.stride(0) for m in chunk]), ia([m.shape[0] for m in chunk]),
             ia([m.shape[1] for m in chunk]), out, ia([p.ld for p in planes]))
        outs += planes
    return outs


def split_planes2(src: torch.Tensor, C: Optional[int] = None, ld: Optional[int] = None, want16: bool = True, wantb: bool = True):
    """fp32 [R, lds] -> (fp16 Planes | None, bf16 Planes | None), both row-major [R][ld], in one pass."""
    R = src.shape[0]
    C = src.shape[1] if C is None else C
    ld = round_up(C, 32) if ld is None else ld
    p16 = Planes.empty(R, ld, torch.float16, src.device, C) if want16 else None
    pb = Planes.empty(R, ld, torch.bfloat16, src.device, C) if wantb else None
    call("hos_split_planes2", ptr(src), src.stride(0), R, C, _pp(p16), ld, _pp(pb), ld)
    return p16, pb


def linearp_fwd(A: Planes, K0: int, W: Planes, bias, M: int, N: int, relu: bool = True, Y: Optional[Planes] = None,
                Yb: Optional[Planes] = None, A1: Optional[Planes] = None, K1: int = 0, C: Optional[torch.Tensor] = None,
                epilogue: int = EPI_NONE, aux=None, aux_col: int = -1, p0: float = 0.0):
    """Y (fp16 planes) / Yb (bf16 planes) = relu?([A | A1] @ W^T + bias), or the fp32 epilogues into C / aux.
    bf16 operands (A.t.dtype == torch.bfloat16): the one-format layer hos_linearp_fwd_b -- the output is Yb (bf16 planes, with its
    ReLU bits), Y must be None."""
    if epilogue == EPI_RESIDUAL:
        aux_col = aux.stride(0)
    if A.t.dtype == torch.bfloat16:
        if Y is not None or W.t.dtype != torch.bfloat16 or (A1 is not None and A1.t.dtype != torch.bfloat16):
            raise _lib.HosLibraryError("linearp_fwd: bf16 operands take bf16 weights and write ONE output format (Yb)")
        _timed(f"gemmp_fwd[M={M},N={N},K={K0 + K1}]", 2.0 * M * N * (K0 + K1), lambda: call(
            "hos_linearp_fwd_b", _pp(A), A.ld, K0, _pp(A1), 0 if A1 is None else A1.ld, K1, _pp(W), W.ld, ptr(bias), M, N, int(relu),
            _pp(Yb), 0 if Yb is None else Yb.ld,
            ptr(Yb.bits, torch.int32) if (relu and Yb is not None and Yb.bits is not None) else 0,
            ptr(C), 0 if C is None else C.stride(0), epilogue, ptr(aux), aux_col, float(p0)))
        return
    _timed(f"gemmp_fwd[M={M},N={N},K={K0 + K1}]", 2.0 * M * N * (K0 + K1), lambda: call(
        "hos_linearp_fwd", _pp(A), A.ld, K0, _pp(A1), 0 if A1 is None else A1.ld, K1, _pp(W), W.ld, ptr(bias), M, N, int(relu),
        _pp(Y), 0 if Y is None else Y.ld, _pp(Yb), 0 if Yb is None else Yb.ld,
        ptr(Y.bits, torch.int32) if (relu and Y is not None and Y.bits is not None) else 0,
        ptr(C), 0 if C is None else C.stride(0), epilogue, ptr(aux), aux_col, float(p0)))


def planes_rowdot(A: Planes, K: int, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, p0: float = 0.0, softplus: bool = True):
    """out[M] = softplus?(A[:, :K] . w[:K] + bias[0] + p0): a one-column head as one pass over the planes A (fp16 or bf16; w: fp32 row
    of the weight, 16-byte aligned; bias: 1-element device tensor or None)."""
    M = A.rows
    fn = "hos_planes_rowdot_b" if A.t.dtype == torch.bfloat16 else "hos_planes_rowdot"
    _timed(f"planes_rowdot[M={M},K={K}]", 2.0 * M * K, lambda: call(
        fn, _pp(A), A.ld, K, ptr(w), ptr(bias), float(p0), int(softplus), M, ptr(out)))
    return out


def linearp_dgrad(dZ: Planes, WT: Planes, Npad: int, M: int, K: int, mask: Optional[Planes] = None,
                  dX: Optional[Planes] = None):
    """dX (bf16 planes [M][ld]) = (dZ @ WT^T) masked by the ReLU bits of `mask` (mask.bits, written by the linearp_fwd that
    produced it) or by mask.hi > 0; WT = transposed weight planes [K][Npad]."""
    bits = None if mask is None else mask.bits
    _timed(f"gemmp_dgrad[M={M},N={K},K={Npad}]", 2.0 * M * K * Npad, lambda: call(
        "hos_linearp_dgrad", _pp(dZ), dZ.ld, _pp(WT), WT.ld, Npad, _pp(None if bits is not None else mask),
        0 if mask is None else mask.ld, ptr(bits, torch.int32), M, K, _pp(dX), dX.ld))


_wgrad_ws = {}          # (device, stream) -> fp32 scratch for the split-K slabs of hos_linearp_wgrad
WGRAD_WS_FLOATS = 16 * 1024 * 1024 + 4096      # 64 MB: 16 slabs of a 1024 x 1024 weight gradient


def _wgrad_workspace(device: torch.device) -> torch.Tensor:
    key = _stream_key(device)
    ws = _wgrad_ws.get(key)
    if ws is None:
        ws = torch.empty(WGRAD_WS_FLOATS, dtype=torch.float32, device=device)
        _wgrad_ws[key] = ws
    return ws


def linearp_wgrad(dZ: Planes, X: Planes, dW: torch.Tensor, db, M: int, N: int, K: int, w_col0: int = 0, splits: int = 0,
                  x_col0: int = 0, use_ws: bool = False):
    """dW[:, w_col0:w_col0+K] += dZ^T @ X[:, x_col0:x_col0+K]; db += column sums of dZ (row-major bf16 planes).
    The split-K partial tiles go through a slab workspace and are summed in a fixed order by default (bit-reproducible and,
    with the eight-loads-per-round reduce kernel, ~1 % faster on the stage-1 step than fp32 atomics); HOS_WGRAD_WS=0 selects
    the atomics."""
    ws = _wgrad_workspace(dW.device) if (use_ws or (WGRAD_WS and N * K >= WGRAD_WS_MIN)) else None
    _timed(f"gemmp_wgrad[M={N},N={K},K={M}]", 2.0 * M * N * K, lambda: call(
        "hos_linearp_wgrad", _pp(dZ), dZ.ld, _pp(X), X.ld, x_col0,
        ptr(dW) + 4 * w_col0, dW.stride(0), ptr(db), M, N, K, splits, ptr(ws), 0 if ws is None else ws.numel()))
