"""CPU: the C-ABI library loads and exports every symbol include/v1t_amd.h declares; host-side logic
(plan / arena layout, registry, error behaviour) without any GPU compute call."""
import ctypes as C
import os
import re
from types import SimpleNamespace

import pytest
import torch

import v1t_amd
from v1t_amd import lib as L
from v1t_amd.synthetic import default_args, make_ds, sensorium_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "v1t_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(v1t_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"libv1t_amd.so does not export {s}"
    assert set(syms) == set(L.SIGNATURES), set(syms) ^ set(L.SIGNATURES)
    assert lib.v1t_abi_version() == 1
    assert lib.v1t_error_string(-2).decode().startswith("configuration not supported")


def test_plan_layout_default_v1t():
    args, ds = sensorium_config({"A": 8000}, input_shape=(1, 36, 64), resize_image=0)
    core = v1t_amd.get_core(args)(args, input_shape=(1, 36, 64))
    assert core.output_shape == (155, 29, 57) and core.num_tokens == 1654 and core.padded_dim == 160
    a = core._arena
    assert a.param_floats == 2465200 == sum(p.numel() for p in core.parameters())
    assert a.total == a.param_floats + 4  # + one `mha.scale` buffer per block
    lib = L.load()
    assert lib.v1t_vit_workspace_bytes(core._plan, 16, 1) > lib.v1t_vit_workspace_bytes(core._plan, 16, 0) > 0
    assert lib.v1t_vit_workspace_offset(core._plan, 16, 1, b"qkv", 1) > lib.v1t_vit_workspace_offset(core._plan, 16, 1, b"qkv", 0) > 0
    assert lib.v1t_vit_workspace_offset(core._plan, 16, 1, b"nope", 0) < 0


def test_arena_views_and_state_dict_roundtrip():
    args, ds = sensorium_config({"A": 100, "B": 37}, input_shape=(1, 36, 64), resize_image=0, num_blocks=2, emb_dim=64, mlp_dim=128)
    m = v1t_amd.Model(args, ds)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.core._arena.ensure()
    arena = m.core._arena
    w = m.core.transformer.blocks[1].mha.to_qkv.weight
    assert w.data_ptr() >= arena.data.data_ptr() and w.grad is not None and w.grad.data_ptr() >= arena.grad.data_ptr()
    ma = m.mouse_arena("B")
    f = m.readouts["B"].features
    assert f.shape == (1, 64, 1, 37) and f.data_ptr() == ma.data.data_ptr()
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # load_state_dict writes through the views into the arena
    sd2 = {k: (v + 1 if v.is_floating_point() else v) for k, v in sd.items()}
    m.load_state_dict(sd2)
    assert torch.equal(m.readouts["B"].feature_storage()[:, :64].t().reshape(1, 64, 1, 37), sd2["readouts.B.features"])
    assert arena.is_current()
    # zero_grad(set_to_none) then re-attach
    for p in m.core.parameters():
        p.grad = None
    arena.grad.fill_(3.0)
    arena.attach_grads()
    assert w.grad is not None and float(w.grad.abs().max()) == 0.0


def test_registry_and_errors():
    args, ds = sensorium_config({"A": 10}, input_shape=(1, 36, 64), resize_image=0, num_blocks=1, emb_dim=64, mlp_dim=128)
    assert v1t_amd.get_core(args) is v1t_amd.ViTCore
    with pytest.raises(NotImplementedError):
        v1t_amd.get_core(SimpleNamespace(core="conv"))  # reference core.py:63-64
    with pytest.raises(NotImplementedError):
        v1t_amd.Readouts(args, model="nope", input_shape=(64, 29, 57), output_shapes=args.output_shapes, ds=ds)
    bad = default_args(input_shape=(1, 36, 64), patch_mode=7)
    bad.output_shapes = {"A": (10,)}
    with pytest.raises(NotImplementedError):
        v1t_amd.ViTCore(bad, input_shape=(1, 36, 64))
    with pytest.raises(ValueError):
        v1t_amd.Gaussian2DReadout(args, input_shape=(64, 29, 57), output_shape=(10,), ds=ds["A"], init_sigma=-1.0)
    # the product path has no CPU fallback: CPU tensors raise RuntimeError
    m = v1t_amd.Model(args, ds)
    x = torch.zeros(1, 1, 36, 64)
    with pytest.raises(RuntimeError):
        m(inputs=x, mouse_id="A", behaviors=torch.zeros(1, 3), pupil_centers=torch.zeros(1, 2))
    plan = C.c_void_p()
    cfg = L.VitConfig(in_channels=1, in_h=36, in_w=64, patch_size=8, patch_stride=1, patch_mode=0, emb_dim=300, num_heads=4, mlp_dim=488,
                      num_blocks=1, behavior_mode=3, num_mice=1, use_lsa=0, use_bias=1, p_dropout=0.0, t_dropout=0.0, ln_eps=1e-5)
    assert L.load().v1t_vit_create(C.byref(cfg), C.byref(plan)) == -2  # head dim > 160: unsupported, not silently wrong


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libv1t_amd.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.load()


def test_product_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "v1t_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("the CPU oracle", "").replace("CPU oracle", ""), f"{f} references the oracle"


def test_mouse_sharding_assignment():
    from v1t_amd.dist import MouseSharding

    ids = list("ABCDEFG")
    for world in (1, 2, 4, 7, 8):
        seen = {m: 0 for m in ids}
        for r in range(world):
            s = MouseSharding(ids, rank=r, world=world, batch_size=16, make_groups=False)
            for m, sl in s.local_units():
                seen[m] += 16 if sl is None else len(range(16)[sl])
        assert all(v == 16 for v in seen.values()), (world, seen)
    s8 = MouseSharding(ids, rank=7, world=8, make_groups=False)
    assert s8.local_units() == [("G", slice(2, 16))] and s8.shared_mice() == ["G"]  # 14 images per rank: the core is batched over pieces
    s81 = MouseSharding(ids, rank=1, world=8, make_groups=False)
    assert s81.local_units() == [("A", slice(14, 16)), ("B", slice(0, 12))] and s81.owners["A"] == [0, 1]
    s2 = MouseSharding(ids, rank=1, world=2, make_groups=False)
    assert s2.local_units() == [("D", slice(8, 16)), ("E", None), ("F", None), ("G", None)]  # 3.5 mice per rank
