"""Shared test helpers (test infrastructure; may import the oracle)."""
from __future__ import annotations

import numpy as np
import torch

MAX_SAMPLE = 4096
MARGINS: dict = {}  # name -> (achieved error, bound): printed by conftest.pytest_terminal_summary, worst first


def record_margin(name: str, err: float, bound: float) -> None:
    old = MARGINS.get(name)
    if old is None or err / max(bound, 1e-300) > old[0] / max(old[1], 1e-300):
        MARGINS[name] = (float(err), float(bound))


def sample(x: torch.Tensor) -> np.ndarray:
    """Same deterministic strided sample as oracle/gen_golden.py."""
    f = x.detach().reshape(-1)
    stride = max(1, -(-f.numel() // MAX_SAMPLE))
    return f[::stride][:MAX_SAMPLE].to(torch.float32).cpu().numpy().copy()


def _np(x):
    if isinstance(x, torch.Tensor):
        return x.detach().to(torch.float64).cpu().numpy()
    return np.asarray(x, dtype=np.float64)


def assert_close(name, got, ref, rtol, atol):
    got = _np(got)
    ref = _np(ref)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = np.abs(got - ref)
    bound = atol + rtol * np.abs(ref)
    worst = float((err / bound).max()) if err.size else 0.0
    record_margin(name, worst, 1.0)
    assert np.isfinite(got).all() and worst <= 1.0, f"{name}: max err {err.max():.3e}, {worst:.2f}x the bound (rtol {rtol}, atol {atol})"


def rel_to_max(got, ref) -> float:
    got = _np(got)
    ref = _np(ref)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))


def check_rel(name: str, got, ref, tol: float) -> float:
    """rel_to_max(got, ref) < tol, with the achieved error / bound recorded for the end-of-run table."""
    e = rel_to_max(got, ref)
    record_margin(name, e, tol)
    assert e < tol, f"{name}: {e:.3e} of the reference's max, bound {tol:.1e} ({e / tol:.2f}x)"
    return e


# ~2-3 x the worst measured over the whole GPU suite of round 6 (1.2e-2 / 6.7e-5: the class-token gradient of the trained-regime step with dropout on;
# everything in the flat regime stays below 4.6e-3 / 1.0e-5)
G_L2_TOL = 2.5e-2  # relative L2 error of a gradient tensor, ||g - ref|| / ||ref||
G_COS_TOL = 2e-4   # 1 - cosine(g, ref)


def check_grad(name: str, got, ref, tol_max: float, tol_l2: float = G_L2_TOL, tol_cos: float = G_COS_TOL) -> float:
    """A gradient tensor against its reference under THREE bounds (a per-tensor max bound alone lets an error that
    is confined to the small entries, or a sign error on ~1 % of the elements, pass): max error relative to the tensor's max (`check_rel`),
    relative L2 error, and 1 - cosine. Margins of all three are recorded for the end-of-run table."""
    e = check_rel(name, got, ref, tol_max)
    g, r = _np(got).reshape(-1), _np(ref).reshape(-1)
    nr = float(np.linalg.norm(r))
    if nr > 0.0:
        l2 = float(np.linalg.norm(g - r)) / nr
        cos = 1.0 - float(np.dot(g, r)) / (float(np.linalg.norm(g)) * nr + 1e-300)
        record_margin(name + " [rel L2]", l2, tol_l2)
        record_margin(name + " [1 - cos]", max(cos, 0.0), tol_cos)
        assert l2 < tol_l2, f"{name}: relative L2 error {l2:.3e}, bound {tol_l2:.1e}"
        assert cos < tol_cos, f"{name}: 1 - cosine {cos:.3e}, bound {tol_cos:.1e}"
    return e


def build_native_model(cfg, sd, device, dropout=None):
    """v1t_amd.Model configured like the oracle Config `cfg`, weights loaded from state-dict `sd`."""
    import v1t_amd
    from v1t_amd.synthetic import default_args, make_ds

    args = default_args(
        input_shape=cfg.raw_input_shape or cfg.input_shape, center_crop=cfg.center_crop, cropper_reg_scale=cfg.cropper_reg_scale,
        shifter_reg_scale=cfg.shifter_reg_scale, readout_reg_scale=cfg.readout_reg_scale, core_reg_scale=cfg.core_reg_scale, resize_image=0, num_blocks=cfg.num_blocks, emb_dim=cfg.emb_dim, mlp_dim=cfg.mlp_dim,
        num_heads=cfg.num_heads, behavior_mode=cfg.behavior_mode, use_lsa=cfg.use_lsa, disable_bias=cfg.disable_bias,
        patch_mode=cfg.patch_mode, patch_stride=cfg.patch_stride, shift_mode=cfg.shift_mode,
        disable_grid_predictor=cfg.disable_grid_predictor, grid_predictor_dim=cfg.grid_predictor_dim,
        p_dropout=cfg.p_dropout, t_dropout=cfg.t_dropout, drop_path=cfg.drop_path, core=cfg.core, pos_emb=cfg.pos_emb,
    )
    args.output_shapes = {m: (cfg.num_neurons[m],) for m in cfg.mouse_ids}
    model = v1t_amd.Model(args, make_ds(cfg.num_neurons))
    r = model.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys and set(r.missing_keys) <= {"image_cropper.grid", "elu1.one"}, r  # every key of the reference-shaped state dict has a home
    return model.to(device), args


def replay_dropout_masks(core, cfg, B, seed, dev):
    """The keep masks the kernels drew for forward seed `seed` (counter-based: `v1t_dropout_mask` evaluates the same hash the
    forward / backward kernels evaluate), shaped for `oracle.v1t_oracle.total_loss(..., masks=...)`: exact-mask replay of the patch,
    attention-P, projection, FC1 and FC2 dropouts (reference vit.py:125-128, 144-151, 229-232, 263)."""
    from v1t_amd import lib as L

    lib = L.load()
    T, D, H, M = core.num_tokens, cfg.emb_dim, cfg.num_heads, cfg.mlp_dim

    def mask(stream, p, rows, cols, shape, take=None):
        m = torch.empty(rows * cols, dtype=torch.uint8, device=dev)
        L.check(lib.v1t_dropout_mask(seed, stream, p, rows, cols, m.data_ptr(), L.stream()))
        m = m.view(rows, cols)
        if take is not None:
            m = m[:, :take]
        return m.reshape(shape).cpu()

    masks = {"patch": mask(0xFFFF, cfg.p_dropout, B * T, core.padded_dim, (B, T, D), D), "attn_p": float(lib.v1t_attention_dropout_rate(cfg.t_dropout))}
    assert abs(masks["attn_p"] - cfg.t_dropout) <= 1 / 512
    for k in range(cfg.num_blocks):
        masks[f"attn{k}"] = mask(8 * k + 0, cfg.t_dropout, B * H * T, T, (B, H, T, T))
        masks[f"proj{k}"] = mask(8 * k + 1, cfg.t_dropout, B * T, core.padded_dim, (B, T, D), D)
        masks[f"fc1{k}"] = mask(8 * k + 2, cfg.t_dropout, B * T, (M + 31) // 32 * 32, (B, T, M), M)
        masks[f"fc2{k}"] = mask(8 * k + 3, cfg.t_dropout, B * T, core.padded_dim, (B, T, D), D)
    return masks


def check_rel_bulk(name: str, got, ref, tol_bulk: float, tol_outlier: float, outlier_frac: float = 1e-3) -> float:
    """Two-level bound for comparisons of the same arithmetic in another launch order (float atomics; one bf16 rounding flip of an
    intermediate, which moves one row of a weight gradient): ALL but a fraction `outlier_frac` of the elements (at least 8) within
    tol_bulk of the reference's max, the remaining few within tol_outlier. A race that drops or doubles a tile of a gradient (128 x 160
    elements) fails the bulk bound, a single rounding flip does not."""
    g, r = _np(got).reshape(-1), _np(ref).reshape(-1)
    assert g.shape == r.shape, (name, g.shape, r.shape)
    err = np.abs(g - r) / (np.abs(r).max() + 1e-30)
    worst = float(err.max()) if err.size else 0.0
    k = min(max(8, int(outlier_frac * err.size)), err.size - 1) if err.size > 1 else 0
    bulk = float(np.partition(err, err.size - 1 - k)[err.size - 1 - k]) if err.size else 0.0  # the largest error once `k` elements are set aside
    record_margin(name + " [bulk]", bulk, tol_bulk)
    record_margin(name + " [outliers]", worst, tol_outlier)
    assert bulk < tol_bulk, f"{name}: all but {k} elements must be within {tol_bulk:.1e} of the reference's max, got {bulk:.3e}"
    assert worst < tol_outlier, f"{name}: worst element {worst:.3e} of the reference's max, bound {tol_outlier:.1e}"
    return worst
