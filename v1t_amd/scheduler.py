"""Checkpointing, learning-rate reduction on plateau and early stopping.

Host-side mirror of src/v1t/utils/scheduler.py:11-198 with the same constructor, `step` / `restore` / `save_checkpoint`
semantics and the same checkpoint file (`<output_dir>/ckpt/model_state.pt` = {"epoch", "value", "model": state_dict,
"optimizer": torch.optim.AdamW-format state, "scheduler": ...}), so a checkpoint of the reference restores into the native
model + fused optimizer and vice versa. No kernels here: the parameters and AdamW moments are views of the flat arenas.
"""
from __future__ import annotations

import os
import typing as t
from collections import OrderedDict

import torch


class Scheduler:
    def __init__(self, args, model, optimizer=None, scaler=None, mode: str = "max", max_reduce: int = 2, lr_patience: int = 10,
                 factor: float = 0.3, min_epochs: int = 0, save_optimizer: bool = True, save_scheduler: bool = True,
                 module_names: t.Optional[t.List[str]] = None):
        assert mode in ("min", "max"), f"mode must be either min or max but not {mode}."
        assert not save_optimizer or optimizer is not None, "Optimizer must be provided when save_optimizer=True"
        if factor >= 1.0:
            raise ValueError("Factor should be < 1.0.")
        self.mode, self.model, self.optimizer, self.scaler, self.module_names = mode, model, optimizer, scaler, module_names
        self.max_reduce, self.num_reduce, self.lr_patience, self.lr_wait = max_reduce, 0, lr_patience, 0
        self.factor, self.min_epochs = factor, min_epochs
        self.best_value = torch.inf if mode == "min" else -torch.inf
        self.checkpoint_dir = os.path.join(args.output_dir, "ckpt")
        os.makedirs(self.checkpoint_dir, exist_ok=True)
        self.save_optimizer, self.save_scheduler = save_optimizer, save_scheduler
        self.device = getattr(args, "device", None)
        self.verbose = getattr(args, "verbose", 0)

    def _parameters2save(self):
        sd = self.model.state_dict()
        if self.module_names is None:
            return sd
        return OrderedDict((k, v) for k, v in sd.items() if k.split(".")[0] in self.module_names)

    def save_checkpoint(self, value, epoch: int):
        filename = os.path.join(self.checkpoint_dir, "model_state.pt")
        ckpt = {"epoch": epoch, "value": float(value), "model": self._parameters2save()}
        if self.save_optimizer:
            ckpt["optimizer"] = self.optimizer.state_dict()
            if self.scaler is not None:
                ckpt["scaler"] = self.scaler.state_dict()
        if self.save_scheduler:
            ckpt["scheduler"] = self.state_dict()
        torch.save(ckpt, f=filename)
        if self.verbose:
            print(f"\nCheckpoint saved to {filename}.")

    def restore(self, force: bool = False, load_optimizer: bool = False, load_scheduler: bool = False) -> int:
        epoch = 0
        filename = os.path.join(self.checkpoint_dir, "model_state.pt")
        if os.path.exists(filename):
            ckpt = torch.load(filename, map_location=self.device, weights_only=False)
            epoch = ckpt["epoch"]
            # the checkpoint may hold only part of the model: update the current state dict (scheduler.py:127-132)
            sd = self.model.state_dict()
            sd.update(ckpt["model"])
            self.model.load_state_dict(sd)
            if load_optimizer and "optimizer" in ckpt:
                self.optimizer.load_state_dict(ckpt["optimizer"])
                if self.scaler is not None and "scaler" in ckpt:
                    self.scaler.load_state_dict(ckpt["scaler"])
            if load_scheduler and "scheduler" in ckpt:
                self.load_state_dict(ckpt["scheduler"])
            if self.verbose:
                print(f"\nLoaded checkpoint from epoch {epoch} (correlation: {ckpt['value']:.04f}).\n")
        elif force:
            raise FileNotFoundError(f"Cannot find checkpoint in {self.checkpoint_dir}.")
        return epoch

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k not in ("optimizer", "model")}

    def load_state_dict(self, state_dict):
        self.__dict__.update(state_dict)

    def is_better(self, value):
        return value < self.best_value if self.mode == "min" else value > self.best_value

    def reduce_lr(self):
        for g in self.optimizer.param_groups:
            g["lr"] = self.factor * float(g["lr"])
            if self.verbose:
                print(f"Reduce learning rate of {g['name']} to {g['lr']:.04e} (num. reduce: {self.num_reduce}).")

    def step(self, value, epoch: int) -> bool:
        terminate = False
        if self.is_better(value):
            self.best_value, self.best_epoch = value, epoch
            self.lr_wait = self.num_reduce = 0
            self.save_checkpoint(value=value, epoch=epoch)
        elif epoch > self.min_epochs:
            if self.lr_wait >= self.lr_patience:
                if self.num_reduce >= self.max_reduce:
                    terminate = True
                    if self.verbose:
                        print(f"\nModel has not improved after {self.num_reduce} LR reductions.")
                else:
                    self.num_reduce += 1
                    self.restore()
                    self.reduce_lr()
                    self.lr_wait = 0
            else:
                self.lr_wait += 1
        return terminate
