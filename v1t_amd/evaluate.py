"""Validation and evaluation loops around the native model.

Host-side mirror of train.py:119-190 (`validation_step`, `validate`) and utils/utils.py:59-199 (`inference`,
`evaluate`) with the same arguments and result dictionaries. The difference is where the predictions go: the reference
moves every micro-batch's predictions to the host and stacks them; here they are folded into device-resident moment
accumulators (`metrics.StreamingMetrics`) as they are produced.
"""
from __future__ import annotations

import typing as t

import numpy as np
import torch

from .metrics import Metrics, StreamingMetrics


def micro_batching(batch: t.Dict[str, t.Any], batch_size: int):
    """reference data.py:106-110"""
    for i in range(0, len(batch["image"]), batch_size):
        yield {k: v[i:i + batch_size] for k, v in batch.items()}


def _num_neurons(model, mouse_id: str) -> int:
    return int(model.output_shapes[mouse_id][0])


@torch.no_grad()
def validation_step(mouse_id: str, batch: t.Dict[str, torch.Tensor], model, criterion, micro_batch_size: int, device: torch.device,
                    metrics: StreamingMetrics) -> t.Dict[str, torch.Tensor]:
    """reference train.py:119-157; predictions go into `metrics` instead of being returned."""
    batch_size = batch["image"].size(0)
    result = {"loss/loss": [], "loss/reg_loss": [], "loss/total_loss": []}
    for mb in micro_batching(batch, micro_batch_size):
        y_true = mb["response"].to(device)
        y_pred, _, _ = model(inputs=mb["image"].to(device), mouse_id=mouse_id, behaviors=mb["behavior"].to(device),
                             pupil_centers=mb["pupil_center"].to(device))
        loss = criterion(y_true=y_true, y_pred=y_pred, mouse_id=mouse_id, batch_size=batch_size)
        reg_loss = (y_true.size(0) / batch_size) * model.regularizer(mouse_id)
        result["loss/loss"].append(loss)
        result["loss/reg_loss"].append(reg_loss)
        result["loss/total_loss"].append(loss + reg_loss)
        metrics.update(y_pred, y_true)
    return {k: torch.sum(torch.stack([torch.as_tensor(x, device=device) for x in v])) for k, v in result.items()}


def log_metrics(results: t.Dict[str, t.Dict[str, t.Any]]) -> t.Dict[str, float]:
    """reference utils/utils.py:349-388 without the TensorBoard side."""
    mouse_ids = list(results.keys())
    names = list(results[mouse_ids[0]].keys())
    for m in mouse_ids:
        for k in names:
            v = results[m][k]
            if isinstance(v, list):
                v = torch.mean(torch.stack(v)) if torch.is_tensor(v[0]) else np.mean(v)
            results[m][k] = float(v)
    return {k[k.find("/") + 1:]: float(np.mean([results[m][k] for m in mouse_ids])) for k in names}


@torch.no_grad()
def validate(args, ds: t.Dict[str, t.Any], model, criterion, epoch: int = 0, sharding=None) -> t.Dict[str, float]:
    """reference train.py:160-190. `sharding`: under mouse-sharded data parallelism the per-mouse modules are current only on
    their owner ranks: they are gathered first (collective: every rank calls this)."""
    if sharding is not None:
        sharding.gather_mice(model)
    model.train(False)
    device = model.device
    mbs = getattr(args, "micro_batch_size", args.batch_size)
    results = {}
    for mouse_id, mouse_ds in ds.items():
        metrics = StreamingMetrics(_num_neurons(model, mouse_id), device)
        mouse_result: t.Dict[str, t.Any] = {}
        for batch in mouse_ds:
            r = validation_step(mouse_id, batch, model, criterion, mbs, device, metrics)
            for k, v in r.items():
                mouse_result.setdefault(k, []).append(v)
        mouse_result["metrics/msse"] = metrics.msse()
        mouse_result["metrics/poisson_loss"] = metrics.poisson_loss()
        mouse_result["metrics/single_trial_correlation"] = metrics.correlation(per_neuron=False)
        results[mouse_id] = mouse_result
    return log_metrics(results)


@torch.no_grad()
def inference(ds, model, micro_batch_size: int, device: torch.device = None) -> t.Dict[str, t.Any]:
    """reference utils/utils.py:59-100; predictions and targets stay in HBM."""
    device = model.device if device is None else device
    results = {"predictions": [], "targets": [], "trial_ids": [], "image_ids": []}
    mouse_id = ds.dataset.mouse_id
    model.train(False)
    for batch in ds:
        for mb in micro_batching(batch, micro_batch_size):
            y, _, _ = model(inputs=mb["image"].to(device), mouse_id=mouse_id, behaviors=mb["behavior"].to(device),
                            pupil_centers=mb["pupil_center"].to(device))
            results["predictions"].append(y)
            results["targets"].append(mb["response"].to(device))
            results["image_ids"].append(mb["image_id"])
            results["trial_ids"].append(mb["trial_id"])
    return {k: torch.cat(v, dim=0) if isinstance(v[0], torch.Tensor) else v for k, v in results.items()}


@torch.no_grad()
def evaluate(args, ds: t.Dict[str, t.Any], model, print_result: bool = False, sharding=None) -> t.Dict[str, float]:
    """reference utils/utils.py:103-199: the three challenge metrics per mouse and their averages. `sharding`: as `validate`."""
    if sharding is not None:
        sharding.gather_mice(model)
    names = ["single_trial_correlation", "correlation_to_average", "feve"]
    results: t.Dict[str, t.Dict[str, float]] = {k: {} for k in names}
    device = model.device
    mbs = getattr(args, "micro_batch_size", args.batch_size)
    model.train(False)
    for mouse_id, mouse_ds in ds.items():
        d = mouse_ds.dataset
        if mouse_id in ("S0", "S1") and d.tier == "test":
            continue
        groups = d.tier == "test" and not d.hashed
        sm = StreamingMetrics(_num_neurons(model, mouse_id), device, image_groups=groups, max_images=len(d) if groups else 0)
        for batch in mouse_ds:
            for mb in micro_batching(batch, mbs):
                y, _, _ = model(inputs=mb["image"].to(device), mouse_id=mouse_id, behaviors=mb["behavior"].to(device),
                                pupil_centers=mb["pupil_center"].to(device))
                sm.update(y, mb["response"], image_ids=mb["image_id"] if groups else None)
        m = Metrics(ds=mouse_ds, results=sm)
        results["single_trial_correlation"][mouse_id] = float(np.mean(m.single_trial_correlation(per_neuron=True)))
        if m.repeat_image and not m.hashed:
            results["correlation_to_average"][mouse_id] = float(np.mean(m.correlation_to_average(per_neuron=True)))
            results["feve"][mouse_id] = float(np.mean(m.feve(per_neuron=True)))
    overall = {}
    for k in names:
        vals = list(results[k].values())
        if vals:
            overall[k] = float(np.mean(vals))
            results[k]["average"] = overall[k]
    if print_result:
        for k in names:
            if results[k]:
                print(k + "\n" + "".join(f"{a}: {b:.04f}\t" for a, b in results[k].items()))
    return overall
