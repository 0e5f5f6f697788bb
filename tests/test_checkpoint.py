"""Checkpoint compatibility (v1t_amd/checkpoint.py, FusedAdamW.state_dict): the file the native side writes has the
structure of the reference's (golden G11: the file utils/scheduler.py:84-104 wrote after the G6 step) and its AdamW moments;
restoring it continues training like the uninterrupted run; optimizer groups follow the reference's order (golden G12)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import v1t_oracle as O
from oracle import weights as W
from tests.helpers import build_native_model, rel_to_max, sample

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_checkpoint_file_roundtrip_cpu(tmp_path):
    """No GPU: file layout, partial (module-filtered) files, atomic replace, missing-file behaviour."""
    from v1t_amd import checkpoint as CK

    class Opt:
        def __init__(self):
            self.param_groups = [{"name": "core", "lr": 1.0}, {"name": "readouts", "lr": 2.0}]
            self.loaded = None

        def state_dict(self):
            return {"state": {}, "param_groups": [dict(g) for g in self.param_groups]}

        def load_state_dict(self, sd):
            self.loaded = sd

    model = torch.nn.ModuleDict({"core": torch.nn.Linear(2, 2), "readouts": torch.nn.Linear(2, 3)})
    opt = Opt()
    assert CK.read(str(tmp_path), model) is None
    with pytest.raises(FileNotFoundError):
        CK.read(str(tmp_path), model, required=True)
    path = CK.write(str(tmp_path), model, opt, epoch=4, value=0.31)
    assert path == str(tmp_path / "ckpt" / "model_state.pt") and os.path.exists(path) and not os.path.exists(path + ".tmp")
    blob = torch.load(path, weights_only=False)
    assert set(blob) == {"epoch", "value", "model", "optimizer", "scheduler"} and blob["scheduler"]["best_value"] == 0.31
    best = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        for p in model.parameters():
            p.add_(1.0)
    info = CK.read(str(tmp_path), model, opt)
    assert info["epoch"] == 4 and info["value"] == 0.31 and opt.loaded["param_groups"][1]["lr"] == 2.0
    assert all(torch.equal(v, best[k]) for k, v in model.state_dict().items())
    # partial file: only the readouts travel; reading it leaves the core as it is
    CK.write(str(tmp_path), model, opt, epoch=5, value=0.4, modules=["readouts"])
    with torch.no_grad():
        for p in model.parameters():
            p.add_(1.0)
    core_now = model["core"].weight.detach().clone()
    CK.read(str(tmp_path), model)
    assert torch.equal(model["core"].weight, core_now) and torch.equal(model["readouts"].weight, best["readouts.weight"])
    other = torch.nn.ModuleDict({"core": torch.nn.Linear(2, 2)})
    with pytest.raises(KeyError):
        CK.read(str(tmp_path), other)


def test_optimizer_groups_follow_reference_order_and_load_by_name():
    """get_parameters() must emit core, readouts, image_cropper, core_shifter like the reference (golden G12,
    model.py:112-139) and FusedAdamW.load_state_dict must match groups by name and check shapes (no GPU needed)."""
    import v1t_amd
    from v1t_amd.synthetic import default_args, make_ds
    from v1t_amd.trainer import FusedAdamW

    g = np.load(os.path.join(GOLD, "g12_boundary.npz"))
    args = default_args(input_shape=(1, 36, 64), center_crop=0.8, resize_image=0, num_blocks=1, emb_dim=64, mlp_dim=128, num_heads=4, shift_mode=4)
    neurons = {"A": 40, "B": 23}
    args.output_shapes = {m: (n,) for m, n in neurons.items()}
    model = v1t_amd.Model(args, make_ds(neurons))
    names = {id(p): k for k, p in model.named_parameters()}
    groups = model.get_parameters(core_lr=1e-3)
    assert [x["name"] for x in groups] == list(g["g12/sm4/group_names"]) == ["core", "readouts", "image_cropper", "core_shifter"]
    for x in groups:
        assert [names[id(p)] for p in x["params"]] == list(g[f"g12/sm4/group/{x['name']}"]), x["name"]
    assert list(model.state_dict().keys()) == list(g["g12/sm4/state_keys"])
    assert tuple(model.core.output_shape) == tuple(g["g12/sm4/core_output_shape"])
    # a reference-ordered optimizer state loads; the same state with two groups swapped (positionally) still loads BY NAME;
    # unnamed + swapped fails on the shape check instead of exchanging moments
    opt = FusedAdamW(1e-3)
    opt.bind(model, 1e-3)
    ref_opt = torch.optim.AdamW(model.get_parameters(core_lr=1e-3), lr=1e-3, betas=(0.9, 0.9999), eps=1e-8, weight_decay=0)
    for p in model.parameters():
        p.grad = torch.full_like(p, 0.5)
    ref_opt.step()
    sd = ref_opt.state_dict()
    opt.load_state_dict(sd)
    a = model.mouse_arena("A")
    assert a.step == 1 and float(a.exp_avg.abs().max()) > 0
    back = opt.state_dict()
    assert [x["name"] for x in back["param_groups"]] == [x["name"] for x in sd["param_groups"]]
    for i, st in sd["state"].items():
        assert torch.allclose(back["state"][i]["exp_avg"], st["exp_avg"]) and back["state"][i]["exp_avg"].shape == st["exp_avg"].shape
    swapped = {"state": sd["state"], "param_groups": [sd["param_groups"][i] for i in (0, 1, 3, 2)]}
    opt.load_state_dict(swapped)
    unnamed = {"state": sd["state"], "param_groups": [{k: v for k, v in x.items() if k != "name"} for x in swapped["param_groups"]]}
    with pytest.raises(ValueError):
        opt.load_state_dict(unnamed)
    # everything is validated BEFORE anything is copied - a file whose parameters of one arena carry different
    # step counts is rejected and leaves moments, step counters and learning rates as they were
    import copy

    bad = copy.deepcopy(sd)
    last = max(bad["state"])
    bad["state"][last]["step"] = torch.tensor(7.0)
    for st in bad["state"].values():
        st["exp_avg"] = st["exp_avg"] + 1.0
    bad["param_groups"][0]["lr"] = 123.0
    before = {m: (model.mouse_arena(m).exp_avg.clone(), model.mouse_arena(m).step) for m in neurons}
    core_before = (model.core._arena.exp_avg.clone(), model.core._arena.step, opt.group_lr("core"))
    with pytest.raises(ValueError, match="step counts"):
        opt.load_state_dict(bad)
    for m in neurons:
        assert torch.equal(model.mouse_arena(m).exp_avg, before[m][0]) and model.mouse_arena(m).step == before[m][1]
    assert torch.equal(model.core._arena.exp_avg, core_before[0]) and model.core._arena.step == core_before[1] and opt.group_lr("core") == core_before[2]


@pytest.mark.gpu
def test_checkpoint_vs_reference_golden_and_resume(tmp_path):
    from v1t_amd import checkpoint as CK
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
    CK.write(str(tmp_path), model, tr.opt, epoch=3, value=0.25, device=dev)
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
    info = CK.read(str(tmp_path), model2, tr2.opt, map_location=dev, required=True)
    assert info["epoch"] == 3 and info["scheduler"]["best_value"] == 0.25
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
