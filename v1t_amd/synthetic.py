"""Synthetic Sensorium-shaped configuration, datasets and batches — no dataset or
checkpoint is available offline, so benchmarks and smoke tests use random-init weights and random data
of the shapes BASELINE.json names. Defaults follow the reference CLI (train.py:328-661)."""
from __future__ import annotations

import typing as t
from types import SimpleNamespace

import numpy as np
import torch

MOUSE_IDS = ("A", "B", "C", "D", "E", "F", "G")


def default_args(**over) -> SimpleNamespace:
    a = SimpleNamespace(
        core="vit", readout="gaussian2d", behavior_mode=3, shift_mode=2, center_crop=1.0, resize_image=1, ds_name="sensorium",
        patch_size=8, patch_mode=0, patch_stride=1, num_blocks=4, num_heads=4, emb_dim=155, mlp_dim=488, p_dropout=0.0229,
        t_dropout=0.2544, drop_path=0.0, use_lsa=False, disable_bias=False, core_reg_scale=0.5379, lr=0.001647, core_lr=None,
        disable_grid_predictor=False, grid_predictor_dim=2, bias_mode=0, readout_reg_scale=0.0076, shifter_reg_scale=0.0,
        cropper_reg_scale=0.0, adam_beta1=0.9, adam_beta2=0.9999, adam_eps=1e-8, batch_size=16, ds_scale=1, seed=1234,
        verbose=0, grad_checkpointing=None, input_shape=(1, 144, 256), output_shapes=None, criterion="poisson",
    )
    for k, v in over.items():
        setattr(a, k, v)
    return a


class SyntheticDataset:
    def __init__(self, mouse_id: str, n: int, size: int, seed: int):
        rng = np.random.default_rng([seed, sum(map(ord, mouse_id))])
        self.mouse_id = mouse_id
        self.coordinates = (rng.standard_normal((n, 3)) * 100.0).astype(np.float32)
        self.response_stats = {"mean": np.abs(rng.standard_normal(n)).astype(np.float32), "std": np.ones(n, np.float32)}
        self.num_neurons = n
        self._size = size

    def __len__(self):
        return self._size


class SyntheticLoader:
    """Quacks like the DataLoader the reference constructors read (`.dataset.coordinates`, `.response_stats`, len)."""

    def __init__(self, mouse_id: str, n: int, size: int = 4500, seed: int = 1234):
        self.dataset = SyntheticDataset(mouse_id, n, size, seed)


def make_ds(neurons: t.Dict[str, int], size: int = 4500, seed: int = 1234) -> t.Dict[str, SyntheticLoader]:
    return {m: SyntheticLoader(m, n, size, seed) for m, n in neurons.items()}


def make_batch(args, mouse_id: str, n: int, batch: int, device, seed: int = 0) -> t.Dict[str, torch.Tensor]:
    g = torch.Generator(device="cpu").manual_seed(seed * 1009 + sum(map(ord, mouse_id)))
    c, h, w = args.input_shape
    out = {
        "image": torch.randn(batch, c, h, w, generator=g),
        "behavior": torch.randn(batch, 3, generator=g).abs(),
        "pupil_center": torch.randn(batch, 2, generator=g),
        "response": torch.empty(batch, n).exponential_(1.0, generator=g),
    }
    return {k: v.to(device) for k, v in out.items()}


def sensorium_config(neurons: t.Optional[t.Dict[str, int]] = None, **over):
    """(args, ds) for BASELINE config C2: default V1T, 7 mice x 8000 neurons, 1x144x256 input."""
    neurons = neurons or {m: 8000 for m in MOUSE_IDS}
    args = default_args(**over)
    args.output_shapes = {m: (n,) for m, n in neurons.items()}
    args.mouse_ids = list(neurons.keys())
    return args, make_ds(neurons, seed=args.seed)
