"""CPU: the oracle (oracle/v1t_oracle.py) against the golden vectors generated from the real reference
(oracle/gen_golden.py, run in the build container). This is what pins the oracle on the GPU box where
/root/reference does not exist."""
import os

import numpy as np
import pytest
import torch

from oracle import v1t_oracle as O
from oracle import weights as W
from tests.helpers import assert_close, sample

RTOL, ATOL = 2e-4, 2e-5  # fp32 re-association between torch's fused kernels and the elementary restatement


def _grads(cfg, sd, batch, mouse, eps=None):
    sdd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    loss, reg, y = O.total_loss(cfg, sdd, batch, mouse, 4500.0, eps=eps)
    (loss + reg).backward()
    return loss.detach(), reg.detach(), y.detach(), sdd


def _cct_b():
    c = W.config_cct({"A": 200})
    c.num_blocks, c.behavior_mode, c.pos_emb, c.emb_dim, c.mlp_dim, c.num_heads = 2, 0, "none", 64, 128, 2
    return c


def _cct_c():
    c = W.config_cct({"A": 200, "B": 123})
    c.num_blocks, c.behavior_mode, c.emb_dim, c.mlp_dim, c.mouse_ids, c.input_shape = 1, 4, 144, 96, ("A", "B"), (2, 36, 64)
    return c


SEEDS = {"g13b": 77, "g13c": 78}


@pytest.mark.parametrize("name,cfg_fn,train", [("g1", W.config_c1, False), ("g4", W.config_c1, True), ("g2b", W.config_c4, False),
                                               ("g13", W.config_cct, False), ("g13b", _cct_b, False), ("g13c", _cct_c, False),
                                               ("g14", lambda: W.config_c2({"A": 8000}), False), ("g14t", lambda: W.config_c2({"A": 8000}), True)])
def test_forward_backward_vs_golden(golden, name, cfg_fn, train):
    cfg = cfg_fn()
    if train:
        cfg.p_dropout = cfg.t_dropout = 0.0
    seed = SEEDS.get(name, 1234)
    sd = (W.make_sharp_state_dict if name.startswith("g14") else W.make_state_dict)(cfg, seed)  # g14*: the trained-weights regime
    batch = W.make_batch(cfg, "A", 2, seed)
    eps = torch.from_numpy(golden[f"{name}/eps"]) if train else None
    loss, reg, y, sdd = _grads(cfg, sd, batch, "A", eps)
    assert_close(f"{name}.y", y.numpy(), golden[f"{name}/y"], RTOL, ATOL)
    assert_close(f"{name}.loss", loss.item(), golden[f"{name}/loss"], RTOL, ATOL)
    assert_close(f"{name}.reg", reg.item(), golden[f"{name}/reg"], RTOL, ATOL)
    n = 0
    for k in golden:
        if k.startswith(f"{name}/grad/"):
            key = k[len(name) + 6 :]
            g = sdd[key].grad
            g = torch.zeros_like(sdd[key]) if g is None else g
            ref = golden[k]
            assert_close(f"{name}.grad.{key}", sample(g), ref, 1e-3, 1e-3 * float(np.abs(ref).max()) + 1e-7)
            n += 1
    assert n >= (16 if name.startswith("g13") else 20)


@pytest.mark.parametrize("name,cfg_fn,seed", [("g1", W.config_c1, 1234), ("dx_patch2", lambda: O.Config(patch_mode=2, **_VAR_BASE), 77),
                                              ("dx_stride2", lambda: O.Config(patch_stride=2, **_VAR_BASE), 77), ("dx_cct", _cct_b, 77)])
def test_input_gradient_vs_golden(golden, name, cfg_fn, seed):
    """d (loss + reg) / d image of the oracle's autograd against the real reference's (golden G14; SURVEY 8(c) G1 "+ core input")."""
    cfg = cfg_fn()
    sd = W.make_state_dict(cfg, seed)
    batch = W.make_batch(cfg, "A", 2, seed)
    batch["image"] = batch["image"].clone().requires_grad_(True)
    loss, reg, _ = O.total_loss(cfg, sd, batch, "A", 4500.0)
    (loss + reg).backward()
    ref = golden[f"{name}/input_grad"]
    assert_close(f"{name}.dx", batch["image"].grad.numpy(), ref, 1e-3, 1e-3 * float(np.abs(ref).max()))


_VAR_BASE = dict(num_blocks=1, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=("A", "B"), num_neurons={"A": 200, "B": 123})


def test_taps_vs_golden(golden):
    cfg = W.config_c1()
    sd = W.make_state_dict(cfg, 1234)
    batch = W.make_batch(cfg, "A", 2, 1234)
    taps = {}
    with torch.no_grad():
        O.model_forward(cfg, sd, batch["image"], "A", batch["behavior"], batch["pupil_center"], taps=taps)
    for k in ("patch_embed", "mha0", "mlp0", "core"):
        assert_close(k, sample(taps[k]), golden[f"g1/tap/{k}"], RTOL, ATOL)


def test_variants_vs_golden(golden):
    base = dict(num_blocks=1, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=("A", "B"), num_neurons={"A": 200, "B": 123})
    variants = {
        "beh0": dict(behavior_mode=0), "beh2": dict(behavior_mode=2), "beh4": dict(behavior_mode=4), "franke": dict(input_shape=(2, 36, 64)),
        "nogridpred": dict(disable_grid_predictor=True), "grid3": dict(grid_predictor_dim=3), "lsa": dict(use_lsa=True),
        "nobias": dict(disable_bias=True), "patch1": dict(patch_mode=1), "patch2": dict(patch_mode=2), "patch3": dict(patch_mode=3),
        "stride2": dict(patch_stride=2), "noshift": dict(shift_mode=0), "heads3_d40": dict(num_heads=3, emb_dim=40, mlp_dim=72),
    }
    for vn, kw in variants.items():
        cfg = O.Config(**{**base, **kw})
        sd = W.make_state_dict(cfg, 77)
        for mouse in ("A", "B"):
            b = W.make_batch(cfg, mouse, 2, 77)
            with torch.no_grad():
                y = O.model_forward(cfg, sd, b["image"], mouse, b["behavior"], b["pupil_center"])
            assert_close(f"variant.{vn}.{mouse}", y.numpy(), golden[f"variant/{vn}/{mouse}/y"], RTOL, ATOL)


def test_rollout_vs_golden(golden):
    cfg = O.Config(num_blocks=2, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=("A",), num_neurons={"A": 64})
    sd = W.make_state_dict(cfg, 99)
    b = W.make_batch(cfg, "A", 2, 99)
    rec = []
    with torch.no_grad():
        O.vit_tokens(cfg, sd, b["image"], "A", b["behavior"], b["pupil_center"], record=rec)
    attn = torch.stack(rec, dim=1)
    assert_close("attn", sample(attn), golden["rollout/attn_sample"], 2e-4, 1e-7)
    for i in range(2):
        assert_close("row", O.attention_rollout_row(attn[i]).numpy(), golden["rollout/row"][i], 1e-4, 1e-9)
        # min-max normalisation amplifies fp32 re-association: absolute tolerance on [0,1]
        assert_close("heat", O.attention_rollout(attn[i], (36, 64)).numpy(), golden["rollout/heatmap"][i], 1e-3, 2e-4)


def test_optimizer_step_vs_golden(golden):
    cfg = O.Config(num_blocks=1, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=("A", "B"), num_neurons={"A": 200, "B": 123}, p_dropout=0.0, t_dropout=0.0)
    sd = W.make_state_dict(cfg, 55)
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    for m in cfg.mouse_ids:
        b = W.make_batch(cfg, m, 4, 55)
        l, r, _ = O.total_loss(cfg, osd, b, m, 4500.0, eps=torch.from_numpy(golden[f"step/eps/{m}"]))
        (l + r).backward()
    keys = O.core_param_keys(sd)
    for m in cfg.mouse_ids:
        keys += O.readout_param_keys(sd, m) + O.shifter_param_keys(sd, m)
    params = {k: osd[k].detach() for k in keys}
    O.adamw_step(params, {k: osd[k].grad for k in keys}, {}, step=1, lr=1.647e-3)
    for k in keys:
        assert_close(f"step.{k}", sample(params[k]), golden[f"step/param/{k}"], 1e-4, 1e-6)


def test_drop_path_vs_golden():
    """G7: stochastic depth in train mode, the reference's torch.rand draws replayed as masks (oracle/gen_golden.py)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g7_drop_path.npz"))
    cfg = O.Config(num_blocks=2, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=("A",), num_neurons={"A": 128}, p_dropout=0.0, t_dropout=0.0, drop_path=0.3)
    sd = W.make_state_dict(cfg, 77)
    batch = W.make_batch(cfg, "A", 6, 77)
    m = torch.from_numpy(g["g7/mask"])
    dpm = {(k, br): m[k, i] for k in range(2) for i, br in enumerate(("mha", "mlp"))}
    sdd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    loss, reg, y = O.total_loss(cfg, sdd, batch, "A", 4500.0, eps=torch.from_numpy(g["g7/eps"]), masks={"drop_path": dpm})
    (loss + reg).backward()
    assert_close("g7.y", y.detach().numpy(), g["g7/y"], 2e-4, 2e-5)
    assert abs(float(loss) - float(g["g7/loss"])) <= 1e-5 * abs(float(g["g7/loss"]))
    for k in ("core.transformer.blocks.0.mha.to_qkv.weight", "core.transformer.blocks.1.mlp.model.4.weight", "core.patch_embedding.pos_embedding"):
        assert_close(f"g7.grad.{k}", sample(sdd[k].grad), g[f"g7/grad/{k}"], 1e-3, 1e-5 + 1e-3 * float(np.abs(g[f"g7/grad/{k}"]).max()))


def _g8_cfg(sm):
    return O.Config(num_blocks=1, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=("A", "B"), num_neurons={"A": 96, "B": 50}, shift_mode=sm,
                    center_crop=0.8, raw_input_shape=(1, 36, 64), input_shape=(1, 28, 51), shifter_reg_scale=0.01, cropper_reg_scale=0.02)


@pytest.mark.parametrize("sm", [1, 3, 4])
def test_image_shifter_vs_golden(sm):
    """G8: center crop 0.8 + the learned image shifter (shift_mode 1/3/4) from the raw image, reference outputs."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g8_image_shift.npz"))
    cfg = _g8_cfg(sm)
    mouse = "B" if sm == 3 else "A"
    sd = W.make_state_dict(cfg, 91)
    batch = W.make_batch(cfg, mouse, 3, 91)
    sdd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    shift = O.image_shifter(cfg, sd, mouse, batch["behavior"], batch["pupil_center"])
    assert_close("g8.shift", shift.numpy(), g[f"g8/sm{sm}/shift"], 1e-5, 1e-7)
    assert np.array_equal(O.crop_nearest(batch["image"], 0.8, shift).numpy(), g[f"g8/sm{sm}/crop"])
    loss, reg, y = O.total_loss(cfg, sdd, batch, mouse, 4500.0)
    (loss + reg).backward()
    assert_close("g8.y", y.detach().numpy(), g[f"g8/sm{sm}/y"], 2e-4, 2e-5)
    assert abs(float(reg) - float(g[f"g8/sm{sm}/reg"])) <= 1e-5 * abs(float(g[f"g8/sm{sm}/reg"]))
    if sm == 4:
        for k in O.image_shifter_param_keys(sd, mouse) + ["core.patch_embedding.pos_embedding"]:
            ref = g[f"g8/sm4/grad/{k}"]
            assert_close(f"g8.grad.{k}", sample(sdd[k].grad), ref, 1e-3, 1e-5 + 1e-3 * float(np.abs(ref).max()))


def test_center_crop_vs_golden():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g8_image_shift.npz"))
    cfg = O.Config(num_blocks=1, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=("A",), num_neurons={"A": 96}, center_crop=0.7,
                   raw_input_shape=(1, 36, 64), input_shape=(1, 25, 44))
    sd = W.make_state_dict(cfg, 91)
    batch = W.make_batch(cfg, "A", 3, 91)
    assert np.array_equal(O.crop_nearest(batch["image"], 0.7, None).numpy(), g["g8/crop07/crop"])
    with torch.no_grad():
        y = O.model_forward_raw(cfg, sd, batch["image"], "A", batch["behavior"], batch["pupil_center"])
    assert_close("crop07.y", y.numpy(), g["g8/crop07/y"], 2e-4, 2e-5)


def test_resize_vs_golden(golden):
    x = torch.from_numpy(np.random.default_rng(5).standard_normal((2, 1, 144, 256)).astype(np.float32))
    assert_close("resize", sample(O.resize_bilinear(x, (36, 64))), golden["resize/out_sample"], 1e-5, 1e-6)


def test_edge_cases():
    """grid points exactly on / beyond +-1 (zero padding after the shift), B = 1."""
    z = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).reshape(2, 3, 4, 5)
    grid = torch.tensor([[[-1.0, -1.0], [1.0, 1.0], [1.5, 0.0], [0.0, -1.2], [1.0 + 1e-7, 1.0]]]).expand(2, -1, -1)
    ref = torch.nn.functional.grid_sample(z, grid[:, :, None, :], align_corners=True)[..., 0]
    assert_close("edges", O.bilinear_sample(z, grid).numpy(), ref.numpy(), 1e-6, 1e-6)
    assert O.find_shape(1653) == (29, 57) and O.find_shape(436 - 1) == (15, 29)


def test_elu1_poisson_edge_vs_golden(golden):
    u = torch.from_numpy(golden["elu_edge/u"]).requires_grad_(True)
    y = torch.from_numpy(golden["elu_edge/y_true"])
    yh = O.elu1(u)
    loss = O.poisson_loss(y, yh, 4500.0, 16)
    loss.backward()
    assert_close("yhat", yh.detach().numpy(), golden["elu_edge/yhat"], 1e-6, 1.2e-7)
    assert_close("loss", loss.item(), golden["elu_edge/loss"], 1e-5, 0)
    assert_close("du", u.grad.numpy(), golden["elu_edge/du"], 1e-6, 1e-12)


def test_metrics_vs_golden():
    """G9: msse / poisson_loss / correlations / FEVe of the real reference on the synthetic test-tier recording."""
    from oracle import metrics_oracle as MO

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g9_metrics.npz"))
    d = MO.make_metric_data()
    cm = MO.compute_metrics(d["targets"], d["predictions"])
    for k, v in cm.items():
        assert abs(v - float(g[f"g9/{k}"])) <= 2e-5 * abs(float(g[f"g9/{k}"])), k
    ot, op, oi = MO.order(d["targets"], d["predictions"], d["image_ids"], d["trial_ids"], d["neuron_ids"])
    assert_close("stc", MO.correlation(op, ot, axis=0), g["g9/single_trial_correlation"], 1e-5, 1e-6)
    assert_close("cta", MO.correlation_to_average(ot, op, oi), g["g9/correlation_to_average"], 1e-5, 1e-6)
    fev, fe = MO.fev_feve(ot, op, oi)
    assert_close("fev", fev, g["g9/fev"], 1e-5, 1e-6)
    assert_close("feve", fe, g["g9/feve"], 1e-5, 1e-6)
    assert_close("feve_kept", MO.feve(ot, op, oi), g["g9/feve_kept"], 1e-5, 1e-6)
