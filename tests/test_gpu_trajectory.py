"""Direct full-size gradient parity and a multi-step trajectory of the step bench.py times (VERDICT r03 "next round" item 7).

  (a) BASELINE configs[1] size (default V1T, T = 1654, D = 155, 8000 neurons), ONE mouse, batch 16, dropout 0, replayed position
      noise: the gradient arenas of the trainer's native step (`_NativeStep`: the fixed C-ABI sequence, no autograd) against the CPU
      oracle's `total_loss(...).backward()` at the same size - no detour through the per-mouse loop of the same HIP kernels
      (reference train.py:42-116).
  (b) 10 optimizer steps at the configs[0] size (1 block / 64-d, 256 neurons, batch 8; dropout 0, replayed noise, a new batch every
      step): loss curve and parameters of the fused trainer (L1 folded into the AdamW kernel) against the oracle's
      loss + regulariser -> backward -> AdamW loop (reference train.py:97-111, 216-223).
"""
import os

import numpy as np
import pytest
import torch

from oracle import v1t_oracle as O
from oracle import weights as W
from tests.helpers import build_native_model, check_grad, check_rel, record_margin

pytestmark = pytest.mark.gpu
G_TOL = 1.2e-2  # the gradient bound of tests/test_gpu_parity.py


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _noop_optimizer(tr):
    tr.opt.step_arena = lambda arena, ranges, zero_grad=True: None  # gradients stay in the arenas, parameters unchanged


@pytest.mark.parametrize("regime", ["flat", "sharp"])
def test_c2_native_step_gradients_vs_oracle_direct(dev, regime):
    from v1t_amd.synthetic import make_ds
    from v1t_amd.trainer import Trainer

    B = 16
    cfg = W.config_c2({"A": 8000})
    cfg.p_dropout = cfg.t_dropout = 0.0
    sd = (W.make_sharp_state_dict if regime == "sharp" else W.make_state_dict)(cfg, 1234)
    batch = W.make_batch(cfg, "A", B, 4321)
    eps = W.make_eps(cfg, "A", B, 4321)
    model, args = build_native_model(cfg, sd, dev)
    args.batch_size = B
    tr = Trainer(args, model, make_ds(cfg.num_neurons))
    assert tr.native
    tr.eps_override = {"A": eps.to(dev)}
    _noop_optimizer(tr)
    out = tr.train_step({"A": {k: v.to(dev) for k, v in batch.items()}})
    torch.cuda.synchronize()
    assert len(tr._native_cache) == 1 and next(iter(tr._native_cache.values())) is not None, "the native step must have run"
    assert model.core._last_ws[1] == B

    # the oracle at the same size on the host's cores (fp32, ~10-60 s)
    nthr = torch.get_num_threads()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    try:
        sdd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
        ol, _, _ = O.total_loss(cfg, sdd, batch, "A", 4500.0, eps=eps)
        ol.backward()
    finally:
        torch.set_num_threads(nthr)
    lo = float(ol)
    errs = []
    record_margin(f"c2 direct [{regime}]: native-step loss vs oracle", abs(float(out["loss"]) - lo), 1e-4 * abs(lo))
    assert abs(float(out["loss"]) - lo) <= 1e-4 * abs(lo)
    n = 0
    for k, p in model.named_parameters():
        ref = sdd[k].grad if k in sdd else None
        if ref is None or p.grad is None:
            continue
        assert bool(torch.isfinite(p.grad).all()), k
        if float(ref.abs().max()) == 0.0:
            assert float(p.grad.abs().max()) == 0.0, k
        else:
            try:
                check_grad(f"c2 direct [{regime}]: native-step grad {k} vs oracle", p.grad.detach().cpu().reshape(ref.shape), ref, G_TOL)
            except AssertionError as ex:
                errs.append(str(ex))
        n += 1
    assert not errs, errs
    assert n >= 60, n  # 4 blocks x 12 + patch embedding + BehaviorMLPs + readout + shifter


REPLAY_CASES = {
    "c2": (lambda: W.config_c2({"A": 8000}), W.make_state_dict),
    # the same step in the regime of trained weights: peaked attention rows, LayerNorm gains 0.3-3, residual outlier channels of +-80,
    # clamped / out-of-range sample positions (oracle/weights.py::make_sharp_state_dict)
    "c2-sharp": (lambda: W.config_c2({"A": 8000}), W.make_sharp_state_dict),
    # BASELINE configs[3] at the bench shape: Franke-shaped 2-channel input, behavior_mode 3, 1121 neurons, batch 16
    "c4": (W.config_c4, W.make_state_dict),
}


@pytest.mark.parametrize("case", sorted(REPLAY_CASES))
def test_c2_native_step_dropout_on_vs_oracle_replayed_masks(dev, case):
    """The step bench.py times, as bench.py runs it: BASELINE configs[1] (4 blocks, D = 155, 4 heads, MLP 488, T = 1654, 8000 neurons),
    one mouse at the metric's batch 16, ALL dropouts ON (p = 0.0229 / 0.2544) and readout sampling on, through `Trainer.train_step` ->
    `_NativeStep` (the fixed C-ABI sequence). The counter-based keep masks of the step's seed are read back through `v1t_dropout_mask`
    and replayed in the CPU oracle together with the position noise: loss and every parameter gradient against
    `total_loss(..., masks=..., eps=...).backward()` (reference vit.py:125-128, 144-151, 229-232, 263; train.py:42-116). This sends the
    replayed masks through attn_fwd<160, dropout>, attn_bwd_dkv2<160, true>, attn_bwd_dq2<160>, ln_gemm<160, ...>, gemm_lnbwd and the
    DP = 160 / MP = 512 dropout epilogues at a 16-image launch."""
    from tests.helpers import replay_dropout_masks
    from v1t_amd.synthetic import make_ds
    from v1t_amd.trainer import Trainer

    B = 16
    cfg_fn, sd_fn = REPLAY_CASES[case]
    cfg = cfg_fn()
    assert cfg.p_dropout > 0 and cfg.t_dropout > 0
    sd = sd_fn(cfg, 1234)
    batch = W.make_batch(cfg, "A", B, 2468)
    eps = W.make_eps(cfg, "A", B, 2468)
    model, args = build_native_model(cfg, sd, dev)
    args.batch_size = B
    tr = Trainer(args, model, make_ds(cfg.num_neurons))
    assert tr.native
    tr.eps_override = {"A": eps.to(dev)}
    _noop_optimizer(tr)
    out = tr.train_step({"A": {k: v.to(dev) for k, v in batch.items()}})
    torch.cuda.synchronize()
    assert len(tr._native_cache) == 1 and next(iter(tr._native_cache.values())) is not None, "the native step must have run"
    core = model.core
    assert core._last_ws[1] == B and core.num_tokens == 1654 and core.padded_dim == 160
    masks = replay_dropout_masks(core, cfg, B, core._seed_state, dev)  # the seed the step's forward / backward used
    keep = float(masks["attn0"].float().mean())
    assert abs(keep - (1.0 - masks["attn_p"])) < 2e-3, keep  # dropout really was on

    nthr = torch.get_num_threads()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    try:
        sdd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
        ol, _, _ = O.total_loss(cfg, sdd, batch, "A", 4500.0, eps=eps, masks=masks)
        ol.backward()
    finally:
        torch.set_num_threads(nthr)
    lo = float(ol)
    errs = []
    record_margin(f"{case} dropout-on: native-step loss vs oracle", abs(float(out["loss"]) - lo), 1e-4 * abs(lo))
    assert abs(float(out["loss"]) - lo) <= 1e-4 * abs(lo)
    n = 0
    for k, p in model.named_parameters():
        ref = sdd[k].grad if k in sdd else None
        if ref is None or p.grad is None:
            continue
        assert bool(torch.isfinite(p.grad).all()), k
        if float(ref.abs().max()) == 0.0:
            assert float(p.grad.abs().max()) == 0.0, k
        else:
            try:
                check_grad(f"{case} dropout-on: native-step grad {k} vs oracle", p.grad.detach().cpu().reshape(ref.shape), ref, G_TOL)
            except AssertionError as ex:  # every tensor is measured (the margins table) before the test fails
                errs.append(str(ex))
        n += 1
    assert not errs, errs
    assert n >= 60, n


def test_c1_ten_step_trajectory_vs_oracle(dev):
    from v1t_amd.synthetic import make_ds
    from v1t_amd.trainer import Trainer

    B, STEPS = 8, 10
    cfg = W.config_c1()
    cfg.p_dropout = cfg.t_dropout = 0.0
    sd = W.make_state_dict(cfg, 77)
    model, args = build_native_model(cfg, sd, dev)
    args.batch_size = B
    lr = float(args.lr)
    tr = Trainer(args, model, make_ds(cfg.num_neurons))
    assert tr.native

    # oracle state: every floating-point entry that the model exposes as a parameter
    pnames = [k for k, _ in model.named_parameters()]
    params = {k: sd[k].clone().requires_grad_(True) for k in pnames}
    osd = dict(sd)
    osd.update(params)
    state: dict = {}
    worst_loss = 0.0
    for s in range(STEPS):
        batch = W.make_batch(cfg, "A", B, 9000 + s)
        eps = W.make_eps(cfg, "A", B, 9000 + s)
        # native
        tr.eps_override = {"A": eps.to(dev)}
        out = tr.train_step({"A": {k: v.to(dev) for k, v in batch.items()}})
        # oracle: loss + regulariser -> backward -> AdamW (train.py:56-72, 216-223)
        for p in params.values():
            p.grad = None
        ol, reg, _ = O.total_loss(cfg, osd, batch, "A", 4500.0, eps=eps)
        (ol + reg).backward()
        grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in params.items()}
        with torch.no_grad():
            O.adamw_step({k: p for k, p in params.items()}, grads, state, s + 1, lr, beta1=args.adam_beta1, beta2=args.adam_beta2, eps=args.adam_eps)
        e = abs(float(out["loss"]) - float(ol)) / abs(float(ol))
        worst_loss = max(worst_loss, e)
        assert e <= 1e-3, f"step {s}: native loss {float(out['loss']):.6e} vs oracle {float(ol):.6e} ({e:.2e})"
    record_margin("c1 trajectory: worst relative loss error over 10 steps", worst_loss, 1e-3)
    assert len(tr._native_cache) == 1 and next(iter(tr._native_cache.values())) is not None, "the native step must have run"
    torch.cuda.synchronize()
    # parameters after 10 steps. Adam moves every element by up to lr per step in the direction of its gradient's sign; where the
    # gradient of an element is at the level of the bf16 operand rounding the sign - and so the direction - is decided by noise, so the
    # bound is on (i) the fraction of elements further than 2 lr from the oracle and (ii) each tensor's mean deviation.
    frac_worst, mean_worst = 0.0, 0.0
    for k, p in model.named_parameters():
        d = (p.detach().cpu().reshape(params[k].shape) - params[k].detach()).abs()
        moved = (params[k].detach() - sd[k]).abs()
        if float(moved.max()) == 0.0:
            assert float(d.max()) == 0.0, k  # parameters without a gradient and without an L1 term stay put in both
            continue
        frac = float((d > 2 * lr).float().mean())
        mean = float(d.mean())
        frac_worst, mean_worst = max(frac_worst, frac), max(mean_worst, mean)
        assert frac <= 0.02, f"{k}: {100 * frac:.2f} % of the elements are further than 2 lr from the oracle after {STEPS} steps"
        assert mean <= 0.5 * lr, f"{k}: mean deviation {mean:.3e} > lr / 2"
    record_margin("c1 trajectory: worst fraction of elements off by > 2 lr", frac_worst, 0.02)
    record_margin("c1 trajectory: worst per-tensor mean deviation", mean_worst, 0.5 * lr)
