"""Checkpoint files interchangeable with the reference's `ckpt/model_state.pt`.

Only the FILE FORMAT is a compatibility surface here; the reference's plateau / early-stopping policy around it
(utils/scheduler.py) is control plane and out of scope. The format, pinned by golden G11 (a file
the reference itself wrote, tests/golden/g11_checkpoint.npz):

    {"epoch": int, "value": float,
     "model": state_dict (reference key names, Appendix C),
     "optimizer": torch.optim.AdamW.state_dict() over model.get_parameters()  (positional parameter ids, named groups),
     "scheduler": {bookkeeping fields the reference restores with `load_scheduler=True`}}

The native side reads / writes it straight from the flat arenas: `FusedAdamW.state_dict()` exposes the AdamW moments as
views of the arenas, so nothing is re-packed. Under mouse-sharded data parallelism the per-mouse arenas (readout, shifters
and their moments) live on their owner ranks only: `write()` first gathers them (`MouseSharding.gather_mice`) so that the
file always holds the complete model, whichever rank writes it.
"""
from __future__ import annotations

import os
import typing as t

import torch

FILENAME = "model_state.pt"

# bookkeeping the reference keeps in the "scheduler" entry (names from golden G11); values are the reference's defaults
# for a run that never reduced its learning rate. A caller with its own policy passes `policy_state` to override them.
_POLICY_DEFAULTS: t.Dict[str, t.Any] = {
    "mode": "max", "max_reduce": 2, "num_reduce": 0, "lr_patience": 10, "lr_wait": 0, "factor": 0.3, "min_epochs": 0,
    "module_names": None, "scaler": None, "save_optimizer": True, "save_scheduler": True, "verbose": 0,
}


def checkpoint_path(output_dir: str) -> str:
    return os.path.join(output_dir, "ckpt", FILENAME)


def write(output_dir: str, model, optimizer, *, epoch: int, value: float, sharding=None, device=None,
          policy_state: t.Optional[t.Dict[str, t.Any]] = None, modules: t.Optional[t.Sequence[str]] = None) -> t.Optional[str]:
    """Write `<output_dir>/ckpt/model_state.pt`. With a multi-rank `sharding` every rank must call this (the gather is
    collective); rank 0 writes and the path is returned there, None elsewhere. `modules`: top-level module names to keep
    (the reference's partial checkpoints), default all."""
    if sharding is not None and sharding.world > 1:
        sharding.gather_mice(model, optimizer)
        if sharding.rank != 0:
            return None
    path = checkpoint_path(output_dir)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    weights = model.state_dict()
    if modules is not None:
        keep = set(modules)
        weights = type(weights)((k, v) for k, v in weights.items() if k.split(".", 1)[0] in keep)
    policy = dict(_POLICY_DEFAULTS)
    policy.update({"best_value": float(value), "best_epoch": int(epoch), "checkpoint_dir": os.path.dirname(path), "device": device,
                   "module_names": None if modules is None else list(modules)})
    policy.update(policy_state or {})
    blob = {"epoch": int(epoch), "value": float(value), "model": weights, "optimizer": optimizer.state_dict(), "scheduler": policy}
    tmp = path + ".tmp"
    torch.save(blob, tmp)
    os.replace(tmp, path)  # a reader never sees a half-written file
    return path


def read(output_dir: str, model, optimizer=None, *, map_location=None, required: bool = False) -> t.Optional[t.Dict[str, t.Any]]:
    """Load a checkpoint (native or written by the reference) into `model` and, if given, the AdamW moments / group
    learning rates into `optimizer`. A partial file (subset of modules) updates only what it holds. Returns
    {"epoch", "value", "scheduler"} or None when there is no file (FileNotFoundError if `required`)."""
    path = checkpoint_path(output_dir)
    if not os.path.exists(path):
        if required:
            raise FileNotFoundError(f"Cannot find checkpoint in {os.path.dirname(path)}.")
        return None
    blob = torch.load(path, map_location=map_location, weights_only=False)
    merged = model.state_dict()
    unknown = [k for k in blob["model"] if k not in merged]
    if unknown:
        raise KeyError(f"checkpoint holds keys the model does not have: {unknown[:5]}")
    merged.update(blob["model"])
    model.load_state_dict(merged)
    if optimizer is not None and "optimizer" in blob:
        optimizer.load_state_dict(blob["optimizer"])
    return {"epoch": int(blob["epoch"]), "value": float(blob["value"]), "scheduler": blob.get("scheduler")}
