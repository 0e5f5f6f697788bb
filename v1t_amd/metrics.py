"""Validation / evaluation metrics computed on the device.

Host-side mirror of the reference's metric code with the same names and results:
- `compute_metrics`  <- train.py:29-39 (msse, poisson_loss, mean single-trial correlation)
- `Metrics`          <- src/v1t/metrics.py:11-142 (single_trial_correlation, correlation_to_average, feve)
- `StreamingMetrics` is what replaces the reference's `vstack(...).cpu()` of every prediction of an epoch
  (train.py:24-25,186; utils/utils.py:93): each micro-batch is folded into per-neuron fp64 moments by the HIP kernels of
  csrc/metrics.hip and only the final per-neuron vectors leave HBM. Row order never matters to these metrics, so the
  reference's re-ordering by trial id (metrics.py:34-44) reduces to the neuron permutation of the outputs.
"""
from __future__ import annotations

import typing as t
from copy import deepcopy

import numpy as np
import torch

from . import lib as L

CORR_EPS = 1e-8   # losses.py:47
LOSS_EPS = 1e-12  # losses.py:35


class StreamingMetrics:
    """Per-neuron moment accumulators for one mouse. `image_groups=True` also keeps the per-image sums that
    correlation_to_average / FEVe need (repeated presentations, test tier)."""

    def __init__(self, num_neurons: int, device: torch.device, image_groups: bool = False, max_images: int = 0):
        self.N, self.device = int(num_neurons), torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("StreamingMetrics runs on the gfx950 kernels only; got device " + str(device))
        self.acc = torch.zeros((5, self.N), dtype=torch.float64, device=self.device)
        self.scal = torch.zeros(2, dtype=torch.float64, device=self.device)
        self.count = 0
        self.G = int(max_images) if image_groups else 0
        self._ids: t.Dict[int, int] = {}
        if self.G:
            self.gacc = torch.zeros((3, self.G, self.N), dtype=torch.float64, device=self.device)
            self.sqerr = torch.zeros(self.N, dtype=torch.float64, device=self.device)
            self.gcount = np.zeros(self.G, dtype=np.int32)

    @staticmethod
    def _f32(x: torch.Tensor, device) -> torch.Tensor:
        return x.to(device=device, dtype=torch.float32).contiguous()

    def update(self, y_pred: torch.Tensor, y_true: torch.Tensor, image_ids: t.Optional[t.Sequence[int]] = None) -> None:
        p, y = self._f32(y_pred, self.device), self._f32(y_true, self.device)
        if p.shape != y.shape or p.dim() != 2 or p.shape[1] != self.N:
            raise RuntimeError(f"StreamingMetrics.update: expected (B, {self.N}) predictions and targets, got {tuple(p.shape)} / {tuple(y.shape)}")
        b = p.shape[0]
        lib = L.load()
        L.check(lib.v1t_metrics_accumulate(p.data_ptr(), y.data_ptr(), b, self.N, LOSS_EPS, self.acc.data_ptr(), self.scal.data_ptr(), L.stream()),
                "metrics_accumulate")
        self.count += b
        if self.G:
            if image_ids is None:
                raise RuntimeError("StreamingMetrics(image_groups=True).update needs image_ids")
            ids = [int(i) for i in (image_ids.tolist() if hasattr(image_ids, "tolist") else image_ids)]
            grp = np.empty(b, dtype=np.int32)
            for i, v in enumerate(ids):
                g = self._ids.setdefault(v, len(self._ids))
                if g >= self.G:
                    raise RuntimeError(f"more than max_images={self.G} distinct image ids")
                grp[i] = g
                self.gcount[g] += 1
            gd = torch.from_numpy(grp).to(self.device)
            L.check(lib.v1t_metrics_group_accumulate(p.data_ptr(), y.data_ptr(), gd.data_ptr(), b, self.N, self.G, self.gacc.data_ptr(),
                                                     self.sqerr.data_ptr(), L.stream()), "metrics_group_accumulate")

    # ------------------------------------------------------------------ results
    def msse(self) -> torch.Tensor:
        return self.scal[0].to(torch.float32)

    def poisson_loss(self) -> torch.Tensor:
        return self.scal[1].to(torch.float32)

    def correlation(self, per_neuron: bool = True) -> torch.Tensor:
        """losses.correlation(y_pred, y_true, dim=0) per neuron, or its mean over neurons."""
        if self.count <= 0:
            raise RuntimeError("no trials accumulated")
        corr = torch.empty(self.N, dtype=torch.float32, device=self.device)
        mean = torch.zeros((), dtype=torch.float64, device=self.device)
        L.check(L.load().v1t_metrics_correlation(self.acc.data_ptr(), self.count, self.N, CORR_EPS, corr.data_ptr(), mean.data_ptr(), L.stream()),
                "metrics_correlation")
        return corr if per_neuron else mean.to(torch.float32)

    def repeat_statistics(self) -> t.Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """(correlation_to_average, fev, feve) per neuron."""
        if not self.G or not self._ids:
            raise RuntimeError("no image groups accumulated")
        g = len(self._ids)
        out = torch.empty((3, self.N), dtype=torch.float32, device=self.device)
        cnt = torch.from_numpy(self.gcount).to(self.device)
        L.check(L.load().v1t_metrics_group_finalize(self.gacc.data_ptr(), cnt.data_ptr(), self.sqerr.data_ptr(), self.G, self.N, CORR_EPS,
                                                    out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), L.stream()), "metrics_group_finalize")
        assert g <= self.G
        return out[0], out[1], out[2]


@torch.no_grad()
def compute_metrics(y_true: torch.Tensor, y_pred: torch.Tensor) -> t.Dict[str, torch.Tensor]:
    """reference train.py:29-39, one launch over device-resident (trials, N) tensors."""
    m = StreamingMetrics(y_pred.shape[1], y_pred.device)
    m.update(y_pred, y_true)
    return {"metrics/msse": m.msse(), "metrics/poisson_loss": m.poisson_loss(), "metrics/single_trial_correlation": m.correlation(per_neuron=False)}


class Metrics:
    """reference src/v1t/metrics.py:11-142 with the same constructor and methods; results are numpy like the
    reference's, the arithmetic runs on the device holding `results["predictions"]`."""

    def __init__(self, ds, results: t.Dict[str, t.Any]):
        self.repeat_image = ds.dataset.tier == "test"
        self.hashed = ds.dataset.hashed
        self.neuron_ids = deepcopy(ds.dataset.neuron_ids)
        self._perm = None if self.hashed else np.argsort(self.neuron_ids)  # Metrics.order, metrics.py:34-44
        if isinstance(results, StreamingMetrics):
            self._m = results
            return
        pred = results["predictions"]
        if not (torch.is_tensor(pred) and pred.is_cuda):
            raise RuntimeError("Metrics: predictions must be a tensor in HBM (the native path has no CPU fallback)")
        image_ids = results["image_ids"]
        image_ids = image_ids.cpu().numpy() if torch.is_tensor(image_ids) else np.asarray(image_ids)
        groups = self.repeat_image and not self.hashed
        self._m = StreamingMetrics(pred.shape[1], pred.device, image_groups=groups, max_images=len(np.unique(image_ids)) if groups else 0)
        self._m.update(pred, results["targets"], image_ids=image_ids if groups else None)

    def _order(self, v: torch.Tensor) -> np.ndarray:
        v = v.cpu().numpy()
        return v if self._perm is None else v[self._perm]

    def single_trial_correlation(self, per_neuron: bool = False):
        corr = self._order(self._m.correlation(per_neuron=True))
        return corr if per_neuron else corr.mean()

    def correlation_to_average(self, per_neuron: bool = False):
        if not self.repeat_image or self.hashed:
            return None
        corr = self._order(self._m.repeat_statistics()[0])
        return corr if per_neuron else corr.mean()

    def feve(self, per_neuron: bool = False, fev_threshold: float = 0.15):
        if not self.repeat_image or self.hashed:
            return None
        _, fev, feve = self._m.repeat_statistics()
        fev, feve = self._order(fev), self._order(feve)
        feve = feve[fev >= fev_threshold]  # ignore neurons below the FEV threshold (metrics.py:140-141)
        return feve if per_neuron else feve.mean()
