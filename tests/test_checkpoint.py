"""Checkpoint compatibility (v1t_amd/scheduler.py, FusedAdamW.state_dict): the file the native Scheduler writes has the
structure of the reference's (golden G11: utils/scheduler.py:84-104 after the G6 step) and its AdamW moments; restoring it
continues training like the uninterrupted run; plateau / early-stopping logic follows scheduler.py:170-198."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import v1t_oracle as O
from oracle import weights as W
from tests.helpers import build_native_model, rel_to_max, sample

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_scheduler_plateau_logic(tmp_path):
    """No GPU: lr reduction after lr_patience epochs without improvement, termination after max_reduce reductions."""
    from v1t_amd.scheduler import Scheduler

    class Opt:
        def __init__(self):
            self.param_groups = [{"name": "core", "lr": 1.0}, {"name": "readouts", "lr": 2.0}]

        def state_dict(self):
            return {"state": {}, "param_groups": [dict(g) for g in self.param_groups]}

        def load_state_dict(self, sd):
            pass

    model = torch.nn.Linear(2, 2)
    sch = Scheduler(SimpleNamespace(output_dir=str(tmp_path), device="cpu", verbose=0), model, Opt(), mode="max", max_reduce=2, lr_patience=1, factor=0.5)
    assert sch.step(0.3, epoch=1) is False and os.path.exists(tmp_path / "ckpt" / "model_state.pt")
    w_best = model.weight.detach().clone()
    with torch.no_grad():
        model.weight.add_(1.0)
    assert sch.step(0.2, epoch=2) is False and sch.lr_wait == 1  # waits
    assert sch.step(0.2, epoch=3) is False and sch.num_reduce == 1  # restores the best weights and reduces
    assert torch.equal(model.weight, w_best) and [g["lr"] for g in sch.optimizer.param_groups] == [0.5, 1.0]
    assert sch.step(0.1, epoch=4) is False and sch.step(0.1, epoch=5) is False and sch.num_reduce == 2
    assert sch.step(0.1, epoch=6) is False and sch.step(0.1, epoch=7) is True  # max_reduce reached
    with pytest.raises(FileNotFoundError):
        Scheduler(SimpleNamespace(output_dir=str(tmp_path / "other"), device="cpu", verbose=0), model, Opt()).restore(force=True)


@pytest.mark.gpu
def test_checkpoint_vs_reference_golden_and_resume(tmp_path):
    from v1t_amd.scheduler import Scheduler
    from v1t_amd.synthetic import make_ds
    from v1t_amd.trainer import Trainer

    dev = torch.device("cuda:0")
    g = np.load(os.path.join(GOLD, "g11_checkpoint.npz"))
    g6 = np.load(os.path.join(GOLD, "g3_g5_g6.npz"))
    cfg = O.Config(num_blocks=1, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=("A", "B"), num_neurons={"A": 200, "B": 123}, p_dropout=0.0, t_dropout=0.0)
    sd = W.make_state_dict(cfg, 55)
    batches = {m: {k: v.to(dev) for k, v in W.make_batch(cfg, m, 4, 55).items()} for m in cfg.mouse_ids}

    def fresh():
        model, args = build_native_model(cfg, sd, dev)
        args.batch_size, args.output_dir, args.device = 4, str(tmp_path), dev
        tr = Trainer(args, model, make_ds(cfg.num_neurons))
        for m in cfg.mouse_ids:  # the reference's eps draws of G6 / G11
            ro, e = model.readouts[m], torch.from_numpy(g6[f"step/eps/{m}"]).to(dev)
            ro.forward = (lambda inputs, sample=None, shifts=None, eps=None, _o=ro.forward, _e=e: _o(inputs, sample=sample, shifts=shifts, eps=_e))
        return model, args, tr

    model, args, tr = fresh()
    tr.train_step(batches)
    sch = Scheduler(args, model, tr.opt, mode="max")
    assert sch.step(0.25, epoch=3) is False
    ck = torch.load(os.path.join(str(tmp_path), "ckpt", "model_state.pt"), weights_only=False)
    # ---- same file structure as the reference's
    assert set(ck) == {"epoch", "value", "model", "optimizer", "scheduler"} and ck["epoch"] == int(g["g11/epoch"]) and ck["value"] == float(g["g11/value"])
    assert set(ck["model"].keys()) == set(g["g11/model_keys"])
    assert sorted(ck["scheduler"].keys()) == sorted(g["g11/scheduler_keys"])
    groups = ck["optimizer"]["param_groups"]
    assert [x["name"] for x in groups] == list(g["g11/group_names"]) and [len(x["params"]) for x in groups] == list(g["g11/group_sizes"])
    np.testing.assert_allclose([x["lr"] for x in groups], g["g11/group_lr"])
    names = {id(p): k for k, p in model.named_parameters()}
    order = [names[id(p)] for grp in tr.opt.param_groups for p in grp["params"]]
    assert order == list(g["g11/opt_param_names"])  # optimizer index -> parameter, as torch numbers them
    worst = 0.0
    for i, k in enumerate(order):
        st = ck["optimizer"]["state"][i]
        assert float(st["step"]) == 1.0 and st["exp_avg"].shape == model.state_dict()[k].shape
        for mom, tol in (("exp_avg", 5e-2), ("exp_avg_sq", 1e-1)):
            ref = g[f"g11/{mom}/{k}"]
            if float(np.abs(ref).max()) > 0:
                e = rel_to_max(sample(st[mom]), ref)
                worst = max(worst, e)
                assert e < tol, (k, mom, e)
    # ---- the optimizer state loads into torch.optim.AdamW over the same parameter groups (what the reference would do)
    topt = torch.optim.AdamW(model.get_parameters(core_lr=args.lr), lr=args.lr, betas=(args.adam_beta1, args.adam_beta2), eps=args.adam_eps, weight_decay=0)
    topt.load_state_dict(ck["optimizer"])
    assert len(topt.state) == len(order)
    # ---- resume: restore into a fresh model + optimizer, then one more step == the uninterrupted run, bit for bit
    tr.train_step(batches)
    cont = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model2, args2, tr2 = fresh()
    sch2 = Scheduler(args2, model2, tr2.opt, mode="max")
    assert sch2.restore(force=True, load_optimizer=True, load_scheduler=True) == 3 and sch2.best_value == 0.25
    tr2.train_step(batches)
    # a few gradients are accumulated with float atomics (LayerNorm gamma / beta, bias column sums), so two runs differ
    # in the last bits of g and Adam's second step by a small fraction of lr where g ~ 0; a run that lost the moments or
    # the step count would be off by ~lr everywhere
    lr = args.lr
    for k, v in model2.state_dict().items():
        if v.is_floating_point() and v.numel() > 1:
            d = (v - cont[k]).abs()
            assert float(d.max()) < 0.25 * lr and float(d.mean()) < 0.01 * lr, (k, float(d.max()), float(d.mean()))
    model3, _, tr3 = fresh()  # control: same weights, NO optimizer state -> visibly different second step
    model3.load_state_dict(ck["model"])
    tr3.train_step(batches)
    k = "core.transformer.blocks.0.mha.to_qkv.weight"
    assert float((model3.state_dict()[k] - cont[k]).abs().mean()) > 0.05 * lr
