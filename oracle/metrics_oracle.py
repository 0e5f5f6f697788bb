"""CPU restatement (numpy, fp32 like the reference) of the reference's validation / evaluation metrics.

TEST INFRASTRUCTURE ONLY: nothing under v1t_amd/ imports this. Pinned against the real reference (`v1t.losses`,
`v1t.metrics.Metrics`) by oracle/gen_golden.py -> tests/golden/g9_metrics.npz.
"""
from __future__ import annotations

import typing as t

import numpy as np


def msse(y_true: np.ndarray, y_pred: np.ndarray) -> np.float32:
    """losses.py:25-29 (reduction "sum")"""
    return np.sum(np.sum(np.square(y_true - y_pred), axis=-1))


def poisson_loss(y_true: np.ndarray, y_pred: np.ndarray, eps: float = 1e-12) -> np.float32:
    """losses.py:32-40 (reduction "sum")"""
    return np.sum(np.sum(y_pred - y_true * np.log(y_pred + np.float32(eps)), axis=-1))


def correlation(y1: np.ndarray, y2: np.ndarray, axis: int = 0, eps: float = 1e-8) -> np.ndarray:
    """losses.py:43-74: standardise each with the population std (+ eps) and average the product."""
    a = (y1 - y1.mean(axis=axis, keepdims=True)) / (y1.std(axis=axis, ddof=0, keepdims=True) + eps)
    b = (y2 - y2.mean(axis=axis, keepdims=True)) / (y2.std(axis=axis, ddof=0, keepdims=True) + eps)
    return (a * b).mean(axis=axis)


def compute_metrics(y_true: np.ndarray, y_pred: np.ndarray) -> t.Dict[str, float]:
    """train.py:29-39"""
    return {"metrics/msse": float(msse(y_true, y_pred)), "metrics/poisson_loss": float(poisson_loss(y_true, y_pred)),
            "metrics/single_trial_correlation": float(correlation(y_pred, y_true, axis=0).mean())}


def order(targets, predictions, image_ids, trial_ids, neuron_ids):
    """Metrics.order metrics.py:34-44: rows by trial id, columns by neuron id."""
    ti, ni = np.argsort(trial_ids), np.argsort(neuron_ids)
    return targets[ti][:, ni], predictions[ti][:, ni], image_ids[ti]


def split_responses(targets, predictions, image_ids):
    """metrics.py:46-63"""
    rt, rp = [], []
    for i in np.unique(image_ids):
        sel = image_ids == i
        rt.append(targets[sel])
        rp.append(predictions[sel])
    return rt, rp


def correlation_to_average(targets, predictions, image_ids) -> np.ndarray:
    """metrics.py:77-93"""
    rt, rp = split_responses(targets, predictions, image_ids)
    mr = np.vstack([x.mean(axis=0, keepdims=True) for x in rt])
    mp = np.vstack([x.mean(axis=0, keepdims=True) for x in rp])
    return correlation(mr, mp, axis=0)


def fev_feve(targets, predictions, image_ids) -> t.Tuple[np.ndarray, np.ndarray]:
    """metrics.py:95-127"""
    rt, rp = split_responses(targets, predictions, image_ids)
    pred_var = np.vstack([(a - b) ** 2 for a, b in zip(rt, rp)])
    img_var = np.vstack([np.var(a, axis=0, ddof=1) for a in rt])
    total_var = np.var(np.vstack(rt), axis=0, ddof=1)
    noise_var = np.mean(img_var, axis=0)
    fev = (total_var - noise_var) / total_var
    feve = 1 - (np.mean(pred_var, axis=0) - noise_var) / (total_var - noise_var)
    return fev, feve


def feve(targets, predictions, image_ids, fev_threshold: float = 0.15) -> np.ndarray:
    """metrics.py:129-142"""
    f, fe = fev_feve(targets, predictions, image_ids)
    return fe[f >= fev_threshold]


def make_metric_data(seed: int = 7, images: int = 20, repeats: int = 6, neurons: int = 257):
    """Synthetic test-tier recording: `images` x `repeats` trials in shuffled order, shuffled neuron ids, responses =
    image signal + trial noise (so FEV spans the 0.15 threshold), predictions = a noisy view of the signal."""
    rng = np.random.default_rng(seed)
    trials = images * repeats
    image_ids = rng.permutation(np.repeat(np.arange(100, 100 + images), repeats)).astype(np.int64)
    trial_ids = rng.permutation(trials).astype(np.int64)
    neuron_ids = rng.permutation(neurons).astype(np.int64) + 1000
    snr = rng.uniform(0.05, 2.0, neurons).astype(np.float32)
    signal = rng.gamma(2.0, 1.0, (images, neurons)).astype(np.float32) * snr
    idx = image_ids - 100
    targets = np.abs(signal[idx] + rng.standard_normal((trials, neurons)).astype(np.float32)).astype(np.float32)
    predictions = np.abs(0.8 * signal[idx] + 0.3 * rng.standard_normal((trials, neurons)).astype(np.float32) + 0.05).astype(np.float32)
    return {"targets": targets, "predictions": predictions, "image_ids": image_ids, "trial_ids": trial_ids, "neuron_ids": neuron_ids}
