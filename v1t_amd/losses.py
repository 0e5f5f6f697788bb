"""Loss / metric pieces on the hot path (reference src/v1t/losses.py).

- `correlation`         <- losses.py:43-89 (the parity metric of BASELINE.json)
- `PoissonLoss`         <- losses.py:141-166 + Loss.scale_ds :114-119 (same call signature)
- `elu1_poisson_loss`   fused HIP op: ELU1 (models/utils.py:109-118) + Poisson loss + dLoss/du in one
                        pass over the (B, N) readout output — the natural epilogue of the readout.
"""
from __future__ import annotations

import math
import typing as t

import torch
from torch import nn

from . import lib as L

EPS = torch.finfo(torch.float32).eps


def correlation(y1: torch.Tensor, y2: torch.Tensor, dim: t.Union[None, int, t.Tuple[int]] = -1, eps: float = 1e-8):
    if dim is None:
        dim = tuple(range(y1.dim()))
    y1 = (y1 - y1.mean(dim=dim, keepdim=True)) / (y1.std(dim=dim, unbiased=False, keepdim=True) + eps)
    y2 = (y2 - y2.mean(dim=dim, keepdim=True)) / (y2.std(dim=dim, unbiased=False, keepdim=True) + eps)
    return (y1 * y2).mean(dim=dim)


class _Elu1PoissonFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, y_true, loss_scale: float):
        L.require_cuda(u, "elu1_poisson_loss")
        u = u.contiguous()
        y_true = y_true.contiguous().to(torch.float32)
        yhat = torch.empty_like(u)
        du = torch.empty_like(u)
        loss = torch.zeros((), dtype=torch.float32, device=u.device)
        L.check(L.load().v1t_elu1_poisson(u.data_ptr(), y_true.data_ptr(), u.numel(), loss_scale, 1.0, yhat.data_ptr(), du.data_ptr(),
                                          loss.data_ptr(), L.stream()), "elu1_poisson")
        ctx.save_for_backward(du)
        ctx.mark_non_differentiable(yhat)
        return loss, yhat

    @staticmethod
    def backward(ctx, gloss, _gy):
        (du,) = ctx.saved_tensors
        return du * gloss, None, None


def elu1_poisson_loss(u: torch.Tensor, y_true: torch.Tensor, ds_size: float, batch_size: int) -> t.Tuple[torch.Tensor, torch.Tensor]:
    """(loss, y_pred): loss = sqrt(ds_size / batch_size) * sum(y_pred + eps - (y_true + eps) * log(y_pred + eps)),
    y_pred = elu(u) + 1."""
    return _Elu1PoissonFn.apply(u, y_true, math.sqrt(ds_size / batch_size))


class _PoissonFn(torch.autograd.Function):
    """The criterion's arithmetic on the model's output as one launch (`v1t_poisson_loss`), dLoss/dy_pred saved for the backward."""

    @staticmethod
    def forward(ctx, y_pred, y_true, eps: float, scale: float):
        y_pred = y_pred.contiguous()
        y_true = y_true.contiguous().to(torch.float32)
        loss = torch.zeros((), dtype=torch.float32, device=y_pred.device)
        dy = torch.empty_like(y_pred) if ctx.needs_input_grad[0] else None
        L.check(L.load().v1t_poisson_loss(y_pred.data_ptr(), y_true.data_ptr(), y_pred.numel(), eps, scale, None if dy is None else dy.data_ptr(),
                                          loss.data_ptr(), L.stream()), "poisson_loss")
        ctx.dy = dy
        return loss

    @staticmethod
    def backward(ctx, g):
        return (None if ctx.dy is None else ctx.dy * g), None, None, None


class PoissonLoss(nn.Module):
    """Same interface as the reference criterion (losses.py:141-166): called on y_pred (post ELU1). fp32 GPU tensors: one fused launch
    forward, one multiply backward (the reference's graph is 7 + 7 small launches per mouse and micro-batch);
    `v1t_amd.install_into_reference()` registers it as the reference's "poisson" criterion."""

    def __init__(self, args, ds: t.Dict[str, t.Any], eps: float = EPS, reduction: str = "sum"):
        super().__init__()
        self.ds_scale = getattr(args, "ds_scale", 1)
        self.ds_sizes = {m: float(len(d.dataset)) for m, d in ds.items()}
        self.register_buffer("eps", torch.tensor(eps))
        self._eps = float(eps)
        self.reduction = reduction

    def forward(self, y_true: torch.Tensor, y_pred: torch.Tensor, mouse_id: str, batch_size: int = None):
        if batch_size is None:
            batch_size = y_true.size(0)
        scale = math.sqrt(self.ds_sizes[mouse_id] / batch_size) if self.ds_scale else 1.0
        if y_pred.is_cuda and y_pred.dtype == torch.float32 and y_true.shape == y_pred.shape:
            return _PoissonFn.apply(y_pred, y_true, self._eps, scale)
        y_true, y_pred = y_true + self.eps, y_pred + self.eps
        loss = torch.sum(y_pred - y_true * torch.log(y_pred))
        if self.ds_scale:
            loss = scale * loss
        return loss
