"""Writes a tiny recording in the Sensorium / Franke on-disk layout (the directory tree src/v1t/data.py:138-227 reads).

TEST INFRASTRUCTURE ONLY (used by oracle/gen_golden.py and the data-path tests); nothing under v1t_amd/ imports it.
"""
from __future__ import annotations

import os

import numpy as np

# data.py:17-41: directory names of the published recordings (only the two used by the fixtures)
DIRS = {("sensorium", "A"): "static21067-10-18-GrayImageNet-94c6ff995dac583098847cfecd43e7b6",
        ("franke2022", "F"): "static25311-10-26-ColorImageNet-104e446ed0128d89c639eef0abe4655b"}


def write_fake_mouse(root: str, ds_name: str, mouse_id: str, seed: int = 0, trials: int = 23, image_shape=(1, 12, 16), neurons: int = 9) -> str:
    rng = np.random.default_rng(seed)
    mouse_dir = os.path.join(root, DIRS[(ds_name, mouse_id)])
    c = image_shape[0]
    if c == 1:  # gray ImageNet frames: integral 0..255
        images = rng.integers(0, 256, size=(trials, *image_shape)).astype(np.float32)
    else:       # colour frames, non-integral
        images = (rng.random((trials, *image_shape)) * 200.0).astype(np.float32)
    responses = rng.gamma(2.0, 3.0, (trials, neurons)).astype(np.float32)
    responses[:, 2] = 0.0  # a silent neuron: std 0 -> the precision falls back to 1 / threshold (data.py:387-397)
    behavior = np.abs(rng.standard_normal((trials, 3))).astype(np.float32) * np.array([10.0, 0.5, 3.0], np.float32)
    pupil = (rng.standard_normal((trials, 2)) * 20 + np.array([150.0, 100.0])).astype(np.float32)
    arrays = {"images": images, "responses": responses, "behavior": behavior, "pupil_center": pupil}
    for d, a in arrays.items():
        os.makedirs(os.path.join(mouse_dir, "data", d), exist_ok=True)
        for i in range(trials):
            np.save(os.path.join(mouse_dir, "data", d, f"{i}.npy"), a[i])
    meta = os.path.join(mouse_dir, "meta")
    for sub in ("neurons", "trials"):
        os.makedirs(os.path.join(meta, sub), exist_ok=True)
    np.save(os.path.join(meta, "neurons", "unit_ids.npy"), rng.permutation(neurons).astype(np.int64) + 1)
    np.save(os.path.join(meta, "neurons", "cell_motor_coordinates.npy"), (rng.standard_normal((neurons, 3)) * 100).astype(np.float64))
    np.save(os.path.join(meta, "neurons", "animal_ids.npy"), np.full(neurons, 21067))
    tiers = np.array(["train", "validation", "test", "train", "train"] * ((trials + 4) // 5))[:trials]
    np.save(os.path.join(meta, "trials", "tiers.npy"), tiers)
    np.save(os.path.join(meta, "trials", "trial_idx.npy"), rng.permutation(trials).astype(np.int64))
    ids = rng.integers(0, 6, trials).astype(np.int64) + 500
    np.save(os.path.join(meta, "trials", "frame_image_id.npy" if ds_name == "sensorium" else "colorframeprojector_image_id.npy"), ids)
    for d, a in arrays.items():
        sd = os.path.join(meta, "statistics", d, "all")
        os.makedirs(sd, exist_ok=True)
        ax = None if d == "images" else 0  # images: one scalar per recording; the rest per column
        stats = {"min": a.min(axis=ax), "max": a.max(axis=ax), "median": np.median(a, axis=ax), "mean": a.mean(axis=ax), "std": a.std(axis=ax)}
        for k, v in stats.items():
            np.save(os.path.join(sd, f"{k}.npy"), np.asarray(v, dtype=np.float32))
    return mouse_dir
