"""
ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Deterministic weights + synthetic inputs shared by the golden-vector generator
(`oracle/gen_golden.py`, runs against the real reference in the build container) and the tests
that replay those vectors on the GPU box. Rule: w[name] = f(seed, name, shape) drawn from
numpy `default_rng`, scaled like the reference initialisers, but with
non-degenerate LayerNorm affine / biases / features so that every term of the math is exercised.
Nothing is copied from the reference: only its state-dict key names and shapes are used.
"""

from __future__ import annotations

import typing as t
import zlib

import numpy as np
import torch

from .v1t_oracle import Config


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def _normal(seed, name, shape, std):
    return (_rng(seed, name).standard_normal(shape) * std).astype(np.float32)


def _uniform(seed, name, shape, lo, hi):
    return _rng(seed, name).uniform(lo, hi, shape).astype(np.float32)


def make_coordinates(seed: int, mouse_id: str, n: int) -> np.ndarray:
    """Synthetic cortical coordinates (N,3) like data.py:199 supplies."""
    return _normal(seed, f"coords.{mouse_id}", (n, 3), 100.0)


def source_grid_from(coords: np.ndarray, dim: int) -> np.ndarray:
    """Normalisation the readout applies to the cortical coordinates (gaussian2d.py:111,133-134)."""
    g = coords[:, :dim].astype(np.float32)
    g = g - g.mean(axis=0, keepdims=True)
    return (g / np.abs(g).max()).astype(np.float32)


def _cct_core_state(cfg: Config, seed: int, put) -> None:
    """Core entries of a CCT model (keys of core/cct.py): conv tokenizer, sine position buffer, blocks with qkv width
    3 * (emb_dim // heads), per-block DropPath buffers."""
    c, _, _ = cfg.input_shape
    D, H, M, P = cfg.emb_dim, cfg.num_heads, cfg.mlp_dim, cfg.patch_size
    inner = D // H
    L = cfg.num_patches
    pd = c * P * P
    put("core.tokenizer.conv2d.weight", _normal(seed, "core.tokenizer.w", (D, c, P, P), np.sqrt(2.0 / pd)))
    if cfg.pos_emb == "sine":  # cct.py:17-27
        pe = np.array([[p / (10000 ** (2 * (i // 2) / D)) for i in range(D)] for p in range(L)], dtype=np.float32)
        pe[:, 0::2] = np.sin(pe[:, 0::2])
        pe[:, 1::2] = np.cos(pe[:, 1::2])
        put("core.tokenizer.pos_embedding", pe[None])
    ws = 0.06
    rates = np.linspace(0, cfg.drop_path, cfg.num_blocks)
    for k in range(cfg.num_blocks):
        b = f"core.transformer.blocks.{k}."
        put(b + "mha.scale", np.float32(inner ** -0.5))
        put(b + "mha.layer_norm.weight", 1.0 + _normal(seed, b + "ln1w", (D,), 0.1))
        put(b + "mha.layer_norm.bias", _normal(seed, b + "ln1b", (D,), 0.1))
        put(b + "mha.qkv.weight", _normal(seed, b + "qkv", (3 * inner, D), 2 * ws))
        put(b + "mha.projection.0.weight", _normal(seed, b + "proj", (D, inner), 2 * ws))
        put(b + "mha.projection.0.bias", _normal(seed, b + "projb", (D,), 0.05))
        put(b + "mlp.0.weight", 1.0 + _normal(seed, b + "ln2w", (D,), 0.1))
        put(b + "mlp.0.bias", _normal(seed, b + "ln2b", (D,), 0.1))
        put(b + "mlp.1.weight", _normal(seed, b + "fc1", (M, D), ws))
        put(b + "mlp.1.bias", _normal(seed, b + "fc1b", (M,), 0.05))
        put(b + "mlp.4.weight", _normal(seed, b + "fc2", (D, M), ws))
        put(b + "mlp.4.bias", _normal(seed, b + "fc2b", (D,), 0.05))
        put(b + "drop_path.keep_prop", np.float32(1.0 - rates[k]))
        if cfg.behavior_mode in (3, 4):
            for key in (cfg.mouse_ids if cfg.behavior_mode == 4 else ("share",)):
                m = f"{b}b_mlp.models.{key}."
                put(m + "0.weight", _normal(seed, m + "w0", (D // 2, 5), 0.3))
                put(m + "0.bias", _normal(seed, m + "b0", (D // 2,), 0.05))
                put(m + "3.weight", _normal(seed, m + "w3", (D, D // 2), 0.15))
                put(m + "3.bias", _normal(seed, m + "b3", (D,), 0.05))
    put("core.reg_scale", np.float32(cfg.core_reg_scale))


def make_state_dict(cfg: Config, seed: int = 1234) -> t.Dict[str, torch.Tensor]:
    c, h, w = cfg.input_shape
    D, H, M, P = cfg.emb_dim, cfg.num_heads, cfg.mlp_dim, cfg.patch_size
    T = cfg.num_patches + 1
    sd: t.Dict[str, np.ndarray] = {}

    def put(k, v):
        sd[k] = np.asarray(v, dtype=np.float32)

    if cfg.core == "cct":
        _cct_core_state(cfg, seed, put)
        return _finish_state_dict(cfg, seed, sd, put)
    return _vit_state_dict(cfg, seed, sd, put, c, D, H, M, P, T)


def _vit_state_dict(cfg, seed, sd, put, c, D, H, M, P, T):

    pe = "core.patch_embedding."
    put(pe + "cls_token", _normal(seed, pe + "cls", (1, 1, D), 1.0))
    put(pe + "pos_embedding", _normal(seed, pe + "pos", (T, D), 1.0))
    if cfg.patch_mode == 0:
        pd = c * P * P
        put(pe + "projection.2.weight", _uniform(seed, pe + "w", (D, pd), -1, 1) / np.sqrt(pd))
        put(pe + "projection.2.bias", _uniform(seed, pe + "b", (D,), -1, 1) / np.sqrt(pd))
    elif cfg.patch_mode == 1:
        pd = c * P * P
        put(pe + "projection.0.weight", _normal(seed, pe + "w", (D, c, P, P), np.sqrt(2.0 / pd)))
        put(pe + "projection.0.bias", _uniform(seed, pe + "b", (D,), -1, 1) / np.sqrt(pd))
    elif cfg.patch_mode == 2:
        pd = (c + 4) * P * P
        put(pe + "projection.3.weight", 1.0 + _normal(seed, pe + "lnw", (pd,), 0.1))
        put(pe + "projection.3.bias", _normal(seed, pe + "lnb", (pd,), 0.1))
        put(pe + "projection.4.weight", _uniform(seed, pe + "w", (D, pd), -1, 1) / np.sqrt(pd))
        put(pe + "projection.4.bias", _uniform(seed, pe + "b", (D,), -1, 1) / np.sqrt(pd))
    elif cfg.patch_mode == 3:
        pd = c * P * P
        put(pe + "projection.2.weight", 1.0 + _normal(seed, pe + "lnw", (pd,), 0.1))
        put(pe + "projection.2.bias", _normal(seed, pe + "lnb", (pd,), 0.1))
        put(pe + "projection.3.weight", _uniform(seed, pe + "w", (D, pd), -1, 1) / np.sqrt(pd))
        put(pe + "projection.3.bias", _uniform(seed, pe + "b", (D,), -1, 1) / np.sqrt(pd))
        put(pe + "projection.4.weight", 1.0 + _normal(seed, pe + "ln2w", (D,), 0.1))
        put(pe + "projection.4.bias", _normal(seed, pe + "ln2b", (D,), 0.1))
    bias = not cfg.disable_bias
    in_dim = 3 if cfg.behavior_mode == 2 else 5
    # weight std: a few times the reference's trunc_normal(0.02) so that attention is not uniform
    ws = 0.06
    for k in range(cfg.num_blocks):
        b = f"core.transformer.blocks.{k}."
        put(b + "mha.layer_norm.weight", 1.0 + _normal(seed, b + "ln1w", (D,), 0.1))
        put(b + "mha.layer_norm.bias", _normal(seed, b + "ln1b", (D,), 0.1))
        put(b + "mha.to_qkv.weight", _normal(seed, b + "qkv", (3 * H * D, D), ws))
        put(b + "mha.projection.0.weight", _normal(seed, b + "proj", (D, H * D), ws))
        if bias:
            put(b + "mha.projection.0.bias", _normal(seed, b + "projb", (D,), 0.05))
        if cfg.use_lsa:
            put(b + "mha.scale", np.full((H,), D**-0.5, np.float32) * (1.0 + _normal(seed, b + "scale", (H,), 0.1)))
            sd[b + "mha.mask"] = np.stack([np.arange(T), np.arange(T)], axis=1).astype(np.int64)
            put(b + "mha.max_value", np.float32(np.finfo(np.float32).max))
        else:
            put(b + "mha.scale", np.float32(D**-0.5))
        put(b + "mlp.model.0.weight", 1.0 + _normal(seed, b + "ln2w", (D,), 0.1))
        put(b + "mlp.model.0.bias", _normal(seed, b + "ln2b", (D,), 0.1))
        put(b + "mlp.model.1.weight", _normal(seed, b + "fc1", (M, D), ws))
        put(b + "mlp.model.4.weight", _normal(seed, b + "fc2", (D, M), ws))
        if bias:
            put(b + "mlp.model.1.bias", _normal(seed, b + "fc1b", (M,), 0.05))
            put(b + "mlp.model.4.bias", _normal(seed, b + "fc2b", (D,), 0.05))
        if cfg.behavior_mode in (2, 3, 4):
            keys = cfg.mouse_ids if cfg.behavior_mode == 4 else ("share",)
            for key in keys:
                m = f"{b}b-mlp.models.{key}."
                put(m + "0.weight", _normal(seed, m + "w0", (D // 2, in_dim), 0.3))
                put(m + "3.weight", _normal(seed, m + "w3", (D, D // 2), 0.15))
                if bias:
                    put(m + "0.bias", _normal(seed, m + "b0", (D // 2,), 0.05))
                    put(m + "3.bias", _normal(seed, m + "b3", (D,), 0.05))
    put("core.reg_scale", np.float32(cfg.core_reg_scale))
    put("core.transformer.drop_path.keep_prop", np.float32(1.0 - cfg.drop_path))
    return _finish_state_dict(cfg, seed, sd, put)


def _finish_state_dict(cfg, seed, sd, put):
    """readouts / shifters (the same modules for every core), then numpy -> torch"""
    D = cfg.emb_dim
    for mid in cfg.mouse_ids:
        n = cfg.num_neurons[mid]
        r = f"readouts.{mid}."
        put(r + "sigma", _uniform(seed, r + "sigma", (1, n, 2, 2), -0.1, 0.1))
        put(r + "features", (1.0 / D) + _normal(seed, r + "feat", (1, D, 1, n), 0.5 / D))
        put(r + "bias", _normal(seed, r + "bias", (n,), 0.3))
        if cfg.disable_grid_predictor:
            put(r + "_mu", _uniform(seed, r + "mu", (1, n, 1, 2), -0.9, 0.9))
        else:
            gd = cfg.grid_predictor_dim
            put(r + "mu_transform.0.weight", _uniform(seed, r + "m0w", (30, gd), -1, 1) * 1.5)
            put(r + "mu_transform.0.bias", _uniform(seed, r + "m0b", (30,), -1, 1) * 0.5)
            put(r + "mu_transform.2.weight", _uniform(seed, r + "m2w", (2, 30), -1, 1) * 0.5)
            put(r + "mu_transform.2.bias", _uniform(seed, r + "m2b", (2,), -1, 1) * 0.2)
            put(r + "source_grid", source_grid_from(make_coordinates(seed, mid, n), gd))
        put(r + "reg_scale", np.float32(cfg.readout_reg_scale))
        if cfg.shift_mode in (2, 3, 4):
            s = f"core_shifter.{mid}.mlp."
            put(s + "0.weight", _uniform(seed, s + "0w", (5, 2), -0.7, 0.7))
            put(s + "0.bias", _uniform(seed, s + "0b", (5,), -0.7, 0.7))
            put(s + "2.weight", _uniform(seed, s + "2w", (5, 5), -0.45, 0.45))
            put(s + "2.bias", _uniform(seed, s + "2b", (5,), -0.45, 0.45))
            put(s + "4.weight", _uniform(seed, s + "4w", (2, 5), -0.45, 0.45) * 0.3)
            put(s + "4.bias", _uniform(seed, s + "4b", (2,), -0.45, 0.45) * 0.3)
            put(f"core_shifter.{mid}.reg_scale", np.float32(cfg.shifter_reg_scale))
        if cfg.shift_mode in (1, 3, 4):  # ImageShifter (image_cropper.py:10-47), num_layers=3, hidden 10
            s = f"image_cropper.image_shifter.{mid}."
            nin = 5 if cfg.shift_mode == 4 else 2
            put(s + "mlp.0.weight", _uniform(seed, s + "0w", (10, nin), -0.7, 0.7))
            put(s + "mlp.0.bias", _uniform(seed, s + "0b", (10,), -0.7, 0.7))
            put(s + "mlp.2.weight", _uniform(seed, s + "2w", (10, 10), -0.45, 0.45))
            put(s + "mlp.2.bias", _uniform(seed, s + "2b", (10,), -0.45, 0.45))
            put(s + "mlp.4.weight", _uniform(seed, s + "4w", (2, 10), -0.45, 0.45))
            put(s + "mlp.4.bias", _uniform(seed, s + "4b", (2,), -0.45, 0.45))
            put(s + "max_shift", np.float32(1.0 - cfg.center_crop))
            put(s + "reg_scale", np.float32(cfg.cropper_reg_scale))
    return {k: torch.from_numpy(np.array(v, copy=True, order="C")) for k, v in sd.items()}


def make_sharp_state_dict(cfg: Config, seed: int = 1234, qk_gain: float = 2.6, outlier: float = 80.0) -> t.Dict[str, torch.Tensor]:
    """`make_state_dict` moved into the regime a TRAINED V1T lives in (every other golden has score std ~ 0.6, i.e.
    nearly flat softmax rows and O(1) activations): the q / k rows of every to_qkv times `qk_gain` (scores ~ gain^2: std 4-6, peaked rows),
    LayerNorm gains spread over [0.3, 3], a handful of residual-stream outlier channels of magnitude `outlier` (through the patch bias and
    one FC2 bias: the "massive activations" of trained transformers, which also shrink every LayerNorm's rstd), readout sigma and core
    shifts large enough that sigma * eps + mu is clamped and mu + shift leaves [-1, 1] (zero-padded taps). ViT cores only."""
    assert cfg.core == "vit"
    sd = {k: v.clone() for k, v in make_state_dict(cfg, seed).items()}
    D, H = cfg.emb_dim, cfg.num_heads
    for k in range(cfg.num_blocks):
        b = f"core.transformer.blocks.{k}."
        sd[b + "mha.to_qkv.weight"][: 2 * H * D] *= qk_gain
        for ln, tag in ((b + "mha.layer_norm.weight", "ln1g"), (b + "mlp.model.0.weight", "ln2g")):
            g = np.exp(_uniform(seed, ln + tag, (D,), np.log(0.3), np.log(3.0)))
            sign = np.where(_rng(seed, ln + tag + "s").uniform(size=D) < 0.1, -1.0, 1.0)
            sd[ln] = torch.from_numpy((g * sign).astype(np.float32))
    # outlier channels: constant over tokens (patch bias) and appearing mid-stream (FC2 bias of block 0); class token shares them via cls
    ch = _rng(seed, "outlier.ch").choice(D, size=4, replace=False)
    amp = np.array([outlier, -0.75 * outlier, 0.6 * outlier, -outlier], np.float32)
    pb = [k for k in sd if k.startswith("core.patch_embedding.projection.") and k.endswith(".bias") and sd[k].shape == (D,)]
    pbk = sorted(pb)[0] if cfg.patch_mode != 3 else "core.patch_embedding.projection.3.bias"
    sd[pbk][ch[:2]] += torch.from_numpy(amp[:2])
    sd["core.patch_embedding.cls_token"][0, 0, ch[:2]] += torch.from_numpy(amp[:2])
    if not cfg.disable_bias:
        sd["core.transformer.blocks.0.mlp.model.4.bias"][ch[2:]] += torch.from_numpy(amp[2:])
    for mid in cfg.mouse_ids:
        n = cfg.num_neurons[mid]
        r = f"readouts.{mid}."
        sd[r + "sigma"] = torch.from_numpy(_uniform(seed, r + "sigma.sharp", (1, n, 2, 2), -0.6, 0.6))
        if cfg.shift_mode in (2, 3, 4):
            s = f"core_shifter.{mid}.mlp."
            sd[s + "4.weight"] = torch.from_numpy(_uniform(seed, s + "4w", (2, 5), -0.45, 0.45) * 2.0)
            sd[s + "4.bias"] = torch.from_numpy(_uniform(seed, s + "4b", (2,), -0.45, 0.45))
        if not cfg.disable_grid_predictor:
            sd[r + "mu_transform.2.weight"] = torch.from_numpy(_uniform(seed, r + "m2w", (2, 30), -1, 1) * 1.5)  # mu = tanh(.) pushed towards +-1
    return sd


def make_batch(cfg: Config, mouse_id: str, batch: int, seed: int = 1234, full_res: bool = False) -> t.Dict[str, torch.Tensor]:
    """Synthetic Sensorium-shaped batch: image ~N(0,1) at the CORE input shape
    (or (C,144,256) pre-cropper when full_res), behavior ~|N(0,1)|, pupil ~N(0,1), response ~Exp(1)."""
    c, h, w = cfg.raw_input_shape or cfg.input_shape
    if full_res:
        h, w = 144, 256
    n = cfg.num_neurons[mouse_id]
    tag = f"batch.{mouse_id}.{batch}"
    return {
        "image": torch.from_numpy(_normal(seed, tag + ".img", (batch, c, h, w), 1.0)),
        "behavior": torch.from_numpy(np.abs(_normal(seed, tag + ".beh", (batch, 3), 1.0))),
        "pupil_center": torch.from_numpy(_normal(seed, tag + ".pup", (batch, 2), 1.0)),
        "response": torch.from_numpy(_rng(seed, tag + ".resp").exponential(1.0, (batch, n)).astype(np.float32)),
    }


def make_eps(cfg: Config, mouse_id: str, batch: int, seed: int = 1234) -> torch.Tensor:
    return torch.from_numpy(_normal(seed, f"eps.{mouse_id}.{batch}", (batch, cfg.num_neurons[mouse_id], 2), 1.0))


# named configurations (BASELINE.json configs)
def config_c1() -> Config:
    """C1: 1-block / 64-d ViT, 1 mouse, 36x64 gray, 256 neurons."""
    return Config(num_blocks=1, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=("A",), num_neurons={"A": 256})


def config_c2(neurons: t.Optional[t.Dict[str, int]] = None) -> Config:
    """C2: default V1T (4 blocks, D=155, 4 heads, MLP 488), 7 mice x ~8k neurons."""
    ids = ("A", "B", "C", "D", "E", "F", "G")
    neurons = neurons or {m: 8000 for m in ids}
    return Config(mouse_ids=tuple(neurons.keys()), num_neurons=dict(neurons))


RAGGED_NEURONS = {"A": 7776, "B": 7939, "C": 8202, "D": 7440, "E": 7928, "F": 8107, "G": 8372}


def config_cct(neurons: t.Optional[t.Dict[str, int]] = None) -> Config:
    """The reference's default CCT arguments (train.py:591-612): 4 blocks, emb_dim 160, 4 heads (qkv width 3 x 40, head dim 10),
    MLP 488, patch 8 / stride 1, sine positions; 36 x 64 gray input -> 18 x 32 = 576 tokens."""
    return Config(core="cct", emb_dim=160, num_neurons=neurons or {"A": 500})


def config_c4() -> Config:
    """C4: Franke-shaped 2-channel input, behavior_mode 3, ~1k neurons."""
    return Config(input_shape=(2, 36, 64), mouse_ids=("A",), num_neurons={"A": 1121})
