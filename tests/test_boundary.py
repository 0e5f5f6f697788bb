"""The drop-in boundary on CPU (no kernels run: constructors, registries, state-dict / optimizer-group contracts).

  * golden G12 (written by the REAL reference's constructors, oracle/gen_golden.py:gen_boundary) pins the native
    Gaussian2DReadout's default initialisation (gaussian2d.py:138-186: sigma ~ U(+-0.1), features = 1/C, bias by bias_mode,
    mu predictor / free mu) under the same torch seed, and the optimizer groups of Model.get_parameters (model.py:112-139);
  * where the reference itself is importable (build container), `install_into_reference()` + the reference's OWN `Model`
    (model.py:74-103) must assemble the native classes with the reference's keys: the north_star's
    "`train.py --core vit --readout gaussian2d` picks them up unchanged".
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import v1t_oracle as O
from oracle import weights as W

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF = "/root/reference/src/v1t"


@pytest.fixture(scope="module")
def g12():
    return np.load(os.path.join(GOLD, "g12_boundary.npz"))


@pytest.mark.parametrize("tag,kw", [("bias0", dict(bias_mode=0)), ("bias1", dict(bias_mode=1)), ("bias2", dict(bias_mode=2)),
                                    ("freemu", dict(bias_mode=0, disable_grid_predictor=True)), ("grid3", dict(bias_mode=0, grid_predictor_dim=3))])
def test_readout_default_init_equals_reference(g12, tag, kw):
    from v1t_amd.readout import Gaussian2DReadout

    n, c = 57, 24
    coords = W.make_coordinates(5, "A", n)
    stats = {"mean": (0.5 + np.arange(n, dtype=np.float32) / n), "std": (1.0 + 0.5 * np.cos(np.arange(n, dtype=np.float32)) ** 2).astype(np.float32)}
    a = SimpleNamespace(readout_reg_scale=0.0076, disable_grid_predictor=False, grid_predictor_dim=2, bias_mode=0)
    for k, v in kw.items():
        setattr(a, k, v)
    ds = SimpleNamespace(dataset=SimpleNamespace(coordinates=coords, response_stats=stats))
    torch.manual_seed(77)
    ro = Gaussian2DReadout(a, input_shape=(c, 5, 7), output_shape=(n,), ds=ds, name="x")
    sd = ro.state_dict()
    want = {k[len(f"g12/init/{tag}/"):]: g12[k] for k in g12.files if k.startswith(f"g12/init/{tag}/")}
    assert set(sd) == set(want)
    for k, ref in want.items():
        got = sd[k].detach().numpy()
        assert got.shape == ref.shape and got.dtype == ref.dtype, k
        # same torch seed, same draw order (predictor Linears, then mu / sigma): bit-identical
        assert np.array_equal(got, ref), (tag, k, float(np.abs(got - ref).max()))
    assert float(sd["features"].min()) == float(sd["features"].max()) == np.float32(1.0 / c)
    assert float(sd["sigma"].abs().max()) <= 0.1 and 0.04 < float(sd["sigma"].abs().mean()) < 0.06  # U(-0.1, 0.1)
    with pytest.raises(NotImplementedError):
        a.bias_mode = 3
        Gaussian2DReadout(a, input_shape=(c, 5, 7), output_shape=(n,), ds=ds, name="x")
    with pytest.raises(ValueError):
        Gaussian2DReadout(a, input_shape=(c, 5, 7), output_shape=(n,), ds=ds, init_sigma=0.0)


def test_optimizer_groups_default_shift_mode(g12):
    from tests.helpers import build_native_model

    cfg = O.Config(num_blocks=1, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=("A", "B"), num_neurons={"A": 40, "B": 23})
    model, _ = build_native_model(cfg, W.make_state_dict(cfg, 3), "cpu")
    names = {id(p): k for k, p in model.named_parameters()}
    groups = model.get_parameters(core_lr=1e-3)
    assert [x["name"] for x in groups] == list(g12["g12/sm2/group_names"])
    for x in groups:
        assert [names[id(p)] for p in x["params"]] == list(g12[f"g12/sm2/group/{x['name']}"])
    assert list(model.state_dict().keys()) == list(g12["g12/sm2/state_keys"])


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference not present")
def test_install_into_reference_builds_native_classes_in_reference_model(g12):
    """The reference's own Model constructor (model.py:74-103), fed by its own registries after install_into_reference()."""
    from oracle import gen_golden as G

    G.import_reference()
    from v1t.models.core import core as ref_core
    from v1t.models.model import Model as RefModel
    from v1t.models.readout import readout as ref_readout

    import v1t_amd

    from v1t import losses as ref_losses

    saved = (ref_core._CORES["vit"], ref_readout._READOUTS["gaussian2d"])
    saved_criterion = ref_losses._CRITERION["poisson"]
    try:
        assert v1t_amd.install_into_reference() is True
        assert ref_core.get_core(SimpleNamespace(core="vit")) is v1t_amd.ViTCore
        # the criterion registry (losses.py:9-17, 193-197): get_criterion() builds the fused PoissonLoss with the reference's own call, and on CPU
        # tensors it computes what the reference's class computes (the fused launch needs fp32 GPU tensors)
        from v1t_amd.losses import PoissonLoss

        ds_c = {"A": SimpleNamespace(dataset=range(4500))}
        crit = ref_losses.get_criterion(SimpleNamespace(criterion="poisson", ds_scale=1, device="cpu"), ds=ds_c)
        assert type(crit) is PoissonLoss
        ref_crit = saved_criterion(SimpleNamespace(ds_scale=1), ds=ds_c)
        gen = torch.Generator().manual_seed(0)
        yt, yp = torch.rand(4, 30, generator=gen) * 3, torch.rand(4, 30, generator=gen) + 0.1
        assert torch.allclose(crit(y_true=yt, y_pred=yp, mouse_id="A", batch_size=4), ref_crit(y_true=yt, y_pred=yp, mouse_id="A", batch_size=4), rtol=1e-6)
        assert torch.allclose(crit(y_true=yt, y_pred=yp, mouse_id="A"), ref_crit(y_true=yt, y_pred=yp, mouse_id="A"), rtol=1e-6)
        for sm, extra in ((2, {}), (4, dict(center_crop=0.8, raw_input_shape=(1, 36, 64), input_shape=(1, 28, 51)))):
            cfg = O.Config(num_blocks=1, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=("A", "B"), num_neurons={"A": 40, "B": 23}, shift_mode=sm, **extra)
            ds = {m: G.FakeDS(W.make_coordinates(3, m, cfg.num_neurons[m]), cfg.num_neurons[m]) for m in cfg.mouse_ids}
            model = RefModel(G.ref_args(cfg), ds=ds)
            assert type(model.core) is v1t_amd.ViTCore
            assert all(type(model.readouts[m]) is v1t_amd.Gaussian2DReadout for m in cfg.mouse_ids)
            assert type(model.readouts).__module__.startswith("v1t.")  # the container and everything around stay the reference's
            assert tuple(model.core.output_shape) == tuple(g12[f"g12/sm{sm}/core_output_shape"])
            assert list(model.state_dict().keys()) == list(g12[f"g12/sm{sm}/state_keys"])
            names = {id(p): k for k, p in model.named_parameters()}
            groups = model.get_parameters(core_lr=1e-3)
            assert [x["name"] for x in groups] == list(g12[f"g12/sm{sm}/group_names"])
            for x in groups:
                assert [names[id(p)] for p in x["params"]] == list(g12[f"g12/sm{sm}/group/{x['name']}"])
            # the reference's OWN Recorder finds what it hooks on the native core (attention_rollout.py:24-36: instances of its Attention
            # class under core.transformer, each with an `.attend` child) - one per block, and they add no state-dict keys (checked above)
            from v1t.models.core import vit as ref_vit
            from v1t.utils.attention_rollout import Recorder as RefRecorder

            rec = RefRecorder(model.core)
            found = rec._find_modules(model.core.transformer, ref_vit.Attention)
            assert len(found) == cfg.num_blocks and all(hasattr(m_, "attend") for m_ in found)
            rec._register_hook()
            assert len(rec.hooks) == cfg.num_blocks and len(model.core._hooked_taps()) == cfg.num_blocks
            assert rec.eject() is model.core and model.core._hooked_taps() == []
            # the opt-in fused optimizer binds to the REFERENCE's Model (which has none of v1t_amd.Model's helper methods): same groups,
            # in the same order, as torch.optim.AdamW over model.get_parameters() (train.py:216-223); zero_grad() attaches every parameter's
            # .grad as a view of its flat gradient arena
            fo = v1t_amd.FusedAdamW.for_model(model, lr=1e-3, core_lr=5e-4)
            to = torch.optim.AdamW(params=model.get_parameters(core_lr=5e-4), lr=1e-3)
            assert [g_["name"] for g_ in fo.param_groups] == [g_["name"] for g_ in to.param_groups]
            assert [len(g_["params"]) for g_ in fo.param_groups] == [len(g_["params"]) for g_ in to.param_groups]
            assert [g_["lr"] for g_ in fo.state_dict()["param_groups"]] == [g_["lr"] for g_ in to.state_dict()["param_groups"]]
            fo.zero_grad()
            assert all(p.grad is not None and float(p.grad.abs().max()) == 0.0 for p in model.parameters() if p.requires_grad)
            # weights written for the reference load into it (same keys and shapes) ...
            res = model.load_state_dict(W.make_state_dict(cfg, 3), strict=False)
            assert not res.unexpected_keys and set(res.missing_keys) <= {"image_cropper.grid", "elu1.one"}
            # ... and the native product path refuses to run on the CPU instead of falling back
            b = W.make_batch(cfg, "A", 2, 3)
            with pytest.raises(RuntimeError):
                model(inputs=b["image"], mouse_id="A", behaviors=b["behavior"], pupil_centers=b["pupil_center"])
    finally:
        ref_core._CORES["vit"], ref_readout._READOUTS["gaussian2d"] = saved
        ref_losses._CRITERION["poisson"] = saved_criterion
        v1t_amd.ViTCore._reference_attention_cls = None
