"""Flat fp32 parameter arenas.

MI355X-first memory layout: all parameters that are updated together live in ONE contiguous HBM
buffer (fp32 master), their gradients in a second buffer of the same layout and the AdamW moments
in two more, so that (i) the HIP backward accumulates straight into the gradient arena (gradient
accumulation over mice, reference train.py:97-111, costs nothing), (ii) the optimizer step is one
fused L1+AdamW launch per arena instead of one per tensor, (iii) the data-parallel exchange is one
RCCL all-reduce over one buffer. nn.Parameters stay ordinary leaves with the reference's shapes and
state-dict names: their `.data` / `.grad` are views into the arenas.
"""
from __future__ import annotations

import typing as t

import torch
from torch import nn


class Slot(t.NamedTuple):
    tensor: torch.Tensor  # nn.Parameter or buffer object (kept by identity)
    offset: int           # floats into the arena
    numel: int            # floats reserved (>= tensor.numel() when the storage is padded)
    view: t.Callable[[torch.Tensor], torch.Tensor]  # storage slice (1-D, numel floats) -> tensor-shaped view
    is_param: bool


class FlatArena:
    """Owns `data`, `grad`, and lazily `exp_avg` / `exp_avg_sq` buffers for a list of slots."""

    def __init__(self, slots: t.Sequence[Slot], total: t.Optional[int] = None, param_floats: t.Optional[int] = None):
        self.slots = list(slots)
        self.total = total if total is not None else sum(s.numel for s in self.slots)
        self.param_floats = param_floats if param_floats is not None else self.total
        self.data: t.Optional[torch.Tensor] = None
        self.grad: t.Optional[torch.Tensor] = None
        self.exp_avg: t.Optional[torch.Tensor] = None
        self.exp_avg_sq: t.Optional[torch.Tensor] = None
        self.step = 0
        self.generation = 0  # bumped on every (re)flatten

    @staticmethod
    def from_params(params: t.Sequence[torch.Tensor], special: t.Optional[t.Dict[int, t.Tuple[int, t.Callable]]] = None) -> "FlatArena":
        """Pack parameters back to back in natural shapes; `special[id(p)] = (numel, view_fn)` overrides
        the storage of one parameter (e.g. neuron-major padded readout features)."""
        slots, off = [], 0
        for p in params:
            if special and id(p) in special:
                n, fn = special[id(p)]
            else:
                n, shape = p.numel(), tuple(p.shape)
                fn = (lambda st, shape=shape: st.view(shape))
            slots.append(Slot(p, off, n, fn, True))
            off += n
        return FlatArena(slots, off, off)

    def is_current(self) -> bool:
        if self.data is None:
            return False
        base = self.data.data_ptr()
        dev = self.data.device
        for s in (self.slots[0], self.slots[-1]) if self.slots else ():
            if s.tensor.device != dev or s.tensor.data_ptr() != base + 4 * s.offset:
                return False
        return True

    def flatten(self) -> None:
        """(Re)build the arenas on the parameters' current device and re-point the parameters."""
        if not self.slots:
            return
        dev = self.slots[0].tensor.device
        old_m, old_v = self.exp_avg, self.exp_avg_sq
        data = torch.zeros(self.total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for s in self.slots:
                s.view(data[s.offset : s.offset + s.numel]).copy_(s.tensor.detach().to(device=dev, dtype=torch.float32))
        self.data = data
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=dev)
        for s in self.slots:
            s.tensor.data = s.view(self.data[s.offset : s.offset + s.numel])
        self.attach_grads(force=True)
        if old_m is not None:
            self.exp_avg, self.exp_avg_sq = old_m.to(dev), old_v.to(dev)
        self.generation += 1

    def ensure(self) -> None:
        if not self.is_current():
            self.flatten()

    def attach_grads(self, force: bool = False) -> None:
        """Point every parameter's .grad at its slice of the gradient arena. After
        `optimizer.zero_grad(set_to_none=True)` the arena is zeroed and the views re-attached."""
        first = next((s for s in self.slots if s.is_param), None)
        if first is None:
            return
        if not force and first.tensor.grad is not None:
            return
        if not force:
            self.grad.zero_()
        for s in self.slots:
            if s.is_param and s.tensor.requires_grad:
                s.tensor.grad = s.view(self.grad[s.offset : s.offset + s.numel])

    def moments(self) -> t.Tuple[torch.Tensor, torch.Tensor]:
        if self.exp_avg is None or self.exp_avg.device != self.data.device:
            self.exp_avg = torch.zeros_like(self.data)
            self.exp_avg_sq = torch.zeros_like(self.data)
        return self.exp_avg, self.exp_avg_sq

    def version(self) -> int:
        return sum(s.tensor._version for s in self.slots)


def init_trunc_normal_(w: torch.Tensor, std: float = 0.02) -> None:
    nn.init.trunc_normal_(w, std=std)
