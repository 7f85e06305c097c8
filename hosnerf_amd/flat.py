"""Flat parameter storage: every parameter of a renderer lives in ONE contiguous fp32 HBM buffer.

Why (MI355X-first): the fused GEMM kernels want zero-padded leading dimensions (multiples of 32
floats) so no kernel has a K-tail path; the optimiser and the gradient all-reduce want one
contiguous buffer (one Adam launch, one RCCL collective over xGMI per step instead of ~50 tiny
ones).  The reference's state_dict names/shapes are kept as *views* into that buffer, so
`load_state_dict` of a reference checkpoint works unchanged (checkpoint row).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class Region:
    """A zero-padded [rows_pad, ld] rectangle inside the flat buffer (1-D: rows_pad == 1)."""

    __slots__ = ("offset", "rows", "cols", "rows_pad", "ld")

    def __init__(self, rows: int, cols: int, rows_pad: Optional[int] = None, ld: Optional[int] = None):
        self.rows, self.cols = rows, cols
        self.rows_pad = rows if rows_pad is None else rows_pad
        self.ld = cols if ld is None else ld
        self.offset = -1

    @property
    def numel(self) -> int:
        return self.rows_pad * self.ld

    def view(self, flat: torch.Tensor) -> torch.Tensor:
        """The whole padded region as a contiguous 2-D tensor [rows_pad, ld]."""
        return flat[self.offset:self.offset + self.numel].view(self.rows_pad, self.ld)


class FlatStore:
    """Allocator + owner of the flat parameter / gradient buffers of one model."""

    ALIGN = 64  # floats (256 B)

    def __init__(self):
        self.regions: List[Region] = []
        self.size = 0
        self.param: Optional[torch.Tensor] = None
        self.grad: Optional[torch.Tensor] = None
        self._bindings: List[Tuple[nn.Parameter, Region, Tuple[slice, ...], Tuple[int, ...]]] = []
        # spans [(offset, numel)] no kernel of a training step reads or writes (see human_nerf.Network: the reference-shaped
        # first deconvolution weight, of which a compact copy is the live one): zero_grad, the norm and Adam skip them
        self.inactive: List[Tuple[int, int]] = []

    def alloc(self, rows: int, cols: int, rows_pad: Optional[int] = None, ld: Optional[int] = None) -> Region:
        r = Region(rows, cols, rows_pad, ld)
        r.offset = self.size
        self.size = _round_up(self.size + r.numel, self.ALIGN)
        self.regions.append(r)
        return r

    def materialize(self, device=None):
        self.param = torch.zeros(self.size, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.size, dtype=torch.float32, device=device)

    # -- parameters as views ---------------------------------------------------------------
    def bind(self, region: Region, index: Tuple[slice, ...], shape: Tuple[int, ...]) -> nn.Parameter:
        """Create an nn.Parameter that is the sub-view region[index].reshape(shape) of the flat buffer."""
        p = nn.Parameter(region.view(self.param)[index].view(shape))
        p.grad = region.view(self.grad)[index].view(shape)
        p._hos_flat_grad = True          # nodes that accumulate into `.grad` in place (ops._DecoderHead) do so only for these
        self._bindings.append((p, region, index, shape))
        return p

    def rebind(self):
        for p, region, index, shape in self._bindings:
            p.data = region.view(self.param)[index].view(shape)
            p.grad = region.view(self.grad)[index].view(shape)

    def to_(self, device):
        if self.param.device == torch.device(device):
            return
        self.param = self.param.to(device)
        self.grad = self.grad.to(device)
        self.rebind()

    def active_spans(self) -> List[Tuple[int, int]]:
        """[(offset, numel)] of the flat buffers a training step touches (everything minus `inactive`)."""
        spans, pos = [], 0
        for off, n in sorted(self.inactive):
            if off > pos:
                spans.append((pos, off - pos))
            pos = max(pos, off + n)
        if pos < self.size:
            spans.append((pos, self.size - pos))
        return spans

    def zero_grad(self):
        self.ensure_bound()
        if self.grad.is_cuda:
            from . import ops          # one library launch for all active spans (the CPU plumbing mode keeps torch's fill)
            ops.copy_or_zero_n([self.grad[off:off + n] for off, n in self.active_spans()])
        elif not self.inactive:
            self.grad.zero_()
        else:
            for off, n in self.active_spans():
                self.grad[off:off + n].zero_()

    def ensure_bound(self):
        """Every parameter's `.grad` must alias the flat gradient buffer: the HIP weight-gradient kernels write into
        `self.grad`, not through autograd.  `Optimizer.zero_grad()` / `nn.Module.zero_grad()` default to set_to_none=True
        and would detach them (the stage would silently stop training) -- re-point any that were dropped or replaced."""
        base, end = self.grad.data_ptr(), self.grad.data_ptr() + 4 * self.grad.numel()
        strays, dirty = [], False
        for p, _, _, _ in self._bindings:
            g = p.grad
            if g is None or not (base <= g.data_ptr() < end):
                dirty = True
                if g is not None and g.shape == p.shape:        # a gradient autograd accumulated outside the flat buffer: keep it
                    strays.append((p, g))
        if dirty:
            self.rebind()
            for p, g in strays:
                p.grad.add_(g)

    def adopt(self, other: "FlatStore") -> int:
        """Append another store's regions to this one (before materialisation); returns the base offset."""
        base = self.size
        self.inactive.extend((o + base, n) for o, n in other.inactive)
        for r in other.regions:
            r.offset += base
            self.regions.append(r)
        self._bindings.extend(other._bindings)
        self.size = _round_up(base + other.size, self.ALIGN)
        return base


class FlatModule(nn.Module):
    """nn.Module whose parameters are views into a FlatStore (`self.store`).

    `.to()/.cuda()/.cpu()` move the flat buffers and re-point every parameter;
    `state_dict()` returns contiguous clones (a view would serialise the whole flat storage).
    """

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "store", FlatStore())
        self._register_state_dict_hook(_clone_views_hook)

    def _apply(self, fn, recurse=True):
        store: FlatStore = self.store
        probe = fn(torch.empty(0, dtype=torch.float32, device=store.param.device))
        if probe.dtype != torch.float32:
            raise TypeError("hosnerf_amd modules are fp32 only (SURVEY 7.1: reduced precision breaks parity)")
        # move the flat buffers first, then let nn.Module move buffers; parameters are re-pointed after
        store.param = fn(store.param)
        store.grad = fn(store.grad)
        out = super()._apply(fn, recurse)
        store.rebind()
        for m in self.modules():
            hook = getattr(m, "_after_flat_move", None)
            if hook is not None:
                hook()
        return out

    def zero_grad(self, set_to_none: bool = False):  # grads alias the flat grad buffer: never drop them
        self.store.zero_grad()

    @property
    def flat_param(self) -> torch.Tensor:
        return self.store.param

    @property
    def flat_grad(self) -> torch.Tensor:
        return self.store.grad


def _clone_views_hook(module, state_dict, prefix, local_metadata):
    for k, v in list(state_dict.items()):
        if isinstance(v, torch.Tensor):
            state_dict[k] = v.detach().clone(memory_format=torch.contiguous_format)
    return state_dict
