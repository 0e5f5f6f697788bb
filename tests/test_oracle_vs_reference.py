"""CPU, build container only: the oracle against the REAL reference imported from /root/reference
(skipped where the reference is absent, e.g. on the GPU box — there tests/golden/ pins the oracle)."""
import os

import pytest
import torch

REF = "/root/reference/src/v1t"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference not present")


def test_c1_forward_backward_matches_reference():
    from oracle import gen_golden as G
    from oracle import v1t_oracle as O
    from oracle import weights as W

    cfg = W.config_c1()
    sd = W.make_state_dict(cfg, 4242)
    batch = W.make_batch(cfg, "A", 3, 4242)
    model = G.build_reference_model(cfg, sd, 4242)
    loss, reg, y, grads, _ = G.ref_forward_backward(model, cfg, batch, "A", 4500.0)
    ol, orr, oy, og = G.oracle_grads(cfg, sd, batch, "A", 4500.0)
    G.check("y", y, oy, 2e-4, 2e-5)
    G.check("loss", loss, ol, 2e-4, 2e-5)
    G.check("reg", reg, orr, 2e-4, 2e-5)
    assert set(grads) == set(og)
    for k, g in grads.items():
        G.check(k, g, og[k], 1e-3, 1e-4 + 2e-4 * float(g.abs().max()))


def test_registry_names_match_reference():
    from oracle import gen_golden as G

    G.import_reference()
    from v1t.models.core.core import _CORES
    from v1t.models.readout.readout import _READOUTS

    import v1t_amd.core as C
    import v1t_amd.readout as R

    assert "vit" in _CORES and "vit" in C._CORES
    assert "gaussian2d" in _READOUTS and "gaussian2d" in R._READOUTS


def test_state_dict_keys_match_reference():
    from oracle import gen_golden as G
    from oracle import v1t_oracle as O
    from oracle import weights as W
    from tests.helpers import build_native_model

    for kw in (dict(), dict(behavior_mode=4), dict(use_lsa=True), dict(disable_bias=True), dict(patch_mode=1), dict(disable_grid_predictor=True), dict(behavior_mode=0, shift_mode=0)):
        cfg = O.Config(**{**dict(num_blocks=2, emb_dim=64, mlp_dim=128, mouse_ids=("A", "B"), num_neurons={"A": 50, "B": 33}), **kw})
        sd = W.make_state_dict(cfg, 1)
        ref = G.build_reference_model(cfg, sd, 1)
        mine, _ = build_native_model(cfg, sd, "cpu")
        rk = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        mk = {k: tuple(v.shape) for k, v in mine.state_dict().items()}
        assert rk == mk, (kw, set(rk) ^ set(mk))
        assert {k for k, _ in ref.named_parameters()} == {k for k, _ in mine.named_parameters()}


def test_cct_state_dict_keys_and_registry_match_reference():
    """The CCT core (core/cct.py): same registry name, same state-dict keys / shapes / parameter set / key ORDER as the reference's
    CCTCore, the sine position table bit-identical to cct.py:17-27, and the same failures for the arguments the reference rejects."""
    from oracle import gen_golden as G
    from oracle import v1t_oracle as O
    from oracle import weights as W
    from tests.helpers import build_native_model

    G.import_reference()
    from v1t.models.core.core import _CORES

    import v1t_amd.core as C

    assert "cct" in _CORES and "cct" in C._CORES
    for kw in (dict(), dict(behavior_mode=4), dict(behavior_mode=0, pos_emb="none"), dict(emb_dim=144, num_heads=4, input_shape=(2, 36, 64)), dict(patch_stride=2)):
        cfg = O.Config(**{**dict(core="cct", num_blocks=2, emb_dim=64, num_heads=2, mlp_dim=128, mouse_ids=("A", "B"), num_neurons={"A": 50, "B": 33}), **kw})
        sd = W.make_state_dict(cfg, 1)
        ref = G.build_reference_model(cfg, sd, 1)
        mine, _ = build_native_model(cfg, sd, "cpu")
        rk = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        mk = {k: tuple(v.shape) for k, v in mine.state_dict().items()}
        assert rk == mk, (kw, set(rk) ^ set(mk))
        assert [k for k in ref.state_dict() if k.startswith("core.")] == [k for k in mine.state_dict() if k.startswith("core.")]
        assert {k for k, _ in ref.named_parameters()} == {k for k, _ in mine.named_parameters()}
        assert tuple(ref.core.output_shape) == tuple(mine.core.output_shape)
    # freshly constructed (not loaded): the position buffer the native constructor computes == the reference's
    import torch

    from v1t.models.core.cct import sinusoidal_embedding as ref_sine
    from v1t_amd.cct import sinusoidal_embedding

    assert torch.equal(sinusoidal_embedding(576, 160), ref_sine(576, 160))
    import pytest as _pt

    bad = O.Config(core="cct", emb_dim=155, num_neurons={"A": 8})  # 155 // 4 = 38 is not divisible by 4: cct.py:111-113 asserts
    with _pt.raises(AssertionError):
        build_native_model(bad, {}, "cpu")
