"""
ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Runs ONLY in the build container.

Generates the golden vectors under tests/golden/ by importing the REAL reference
(/root/reference/src/v1t, read-only, never copied) with stub modules for its missing third-party
imports, loading the deterministic weights of `oracle/weights.py` through
`load_state_dict`, and running its `Model` / `Recorder` / `attention_rollouts` on the synthetic
inputs of the same module. While generating, every vector is also compared against the CPU
restatement `oracle/v1t_oracle.py` in fp32 and fp64 — the script aborts if they disagree, which
is what pins the oracle to the reference.

    python -m oracle.gen_golden            # writes tests/golden/*.npz

Fixtures hold data only (inputs are regenerated from seeds; expected outputs / gradient samples
are stored). Large tensors are stored as deterministic strided samples (`sample()` below).
"""

from __future__ import annotations

import os
import sys
import types
import typing as t
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import v1t_oracle as O  # noqa: E402
from oracle import weights as W  # noqa: E402

MAX_SAMPLE = 4096


def sample(x: torch.Tensor) -> np.ndarray:
    """Deterministic strided sample of a tensor: flat[::stride][:MAX_SAMPLE]."""
    f = x.detach().reshape(-1)
    stride = max(1, -(-f.numel() // MAX_SAMPLE))
    return f[::stride][:MAX_SAMPLE].to(torch.float32).numpy().copy()


# --------------------------------------------------------------------------------------
# reference import with stubs
# --------------------------------------------------------------------------------------
def import_reference():
    if "v1t" in sys.modules:
        return
    sys.path.insert(0, os.path.join(REF, "src"))

    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    stub("torchinfo", summary=lambda *a, **k: None)
    stub("wandb")
    ru = stub("ruamel")  # v1t.utils.yaml (imported by v1t.utils.utils, which v1t.data imports); never called here
    ru.yaml = stub("ruamel.yaml", YAML=lambda *a, **k: SimpleNamespace())

    class Resize:
        def __init__(self, size, antialias=False):
            self.size = size

        def __call__(self, x):
            return F.interpolate(x, size=self.size, mode="bilinear", align_corners=False, antialias=False)

    def resize(img, size, antialias=False):
        return F.interpolate(img[None], size=tuple(size), mode="bilinear", align_corners=False, antialias=False)[0]

    tv = stub("torchvision")
    tvt = stub("torchvision.transforms", Resize=Resize)
    tvf = stub("torchvision.transforms.functional", resize=resize)
    tv.transforms = tvt
    tvt.functional = tvf
    tb = stub("v1t.utils.tensorboard", Summary=object)
    import v1t.utils  # noqa: F401

    sys.modules["v1t.utils"].tensorboard = tb


class FakeDS:
    """What the reference constructors read from a DataLoader (readout.py:36, gaussian2d.py:186, losses.py:107-111)."""

    def __init__(self, coords: np.ndarray, n: int, size: int = 4500):
        self.dataset = SimpleNamespace(
            coordinates=coords,
            response_stats={"mean": np.abs(np.ones(n, np.float32)), "std": np.ones(n, np.float32)},
        )
        self._size = size


def ref_args(cfg: O.Config, ds_name="sensorium"):
    return SimpleNamespace(
        core=cfg.core, pos_emb=cfg.pos_emb, readout="gaussian2d", behavior_mode=cfg.behavior_mode, shift_mode=cfg.shift_mode,
        center_crop=cfg.center_crop, resize_image=0, ds_name=ds_name, patch_size=cfg.patch_size, patch_mode=cfg.patch_mode,
        patch_stride=cfg.patch_stride, num_blocks=cfg.num_blocks, num_heads=cfg.num_heads, emb_dim=cfg.emb_dim,
        mlp_dim=cfg.mlp_dim, p_dropout=cfg.p_dropout, t_dropout=cfg.t_dropout, drop_path=cfg.drop_path, use_lsa=cfg.use_lsa,
        disable_bias=cfg.disable_bias, core_reg_scale=cfg.core_reg_scale,
        disable_grid_predictor=cfg.disable_grid_predictor, grid_predictor_dim=cfg.grid_predictor_dim,
        bias_mode=cfg.bias_mode, readout_reg_scale=cfg.readout_reg_scale, shifter_reg_scale=cfg.shifter_reg_scale,
        cropper_reg_scale=cfg.cropper_reg_scale, device=torch.device("cpu"), verbose=0, grad_checkpointing=0,
        input_shape=cfg.raw_input_shape or cfg.input_shape, output_shapes={m: (cfg.num_neurons[m],) for m in cfg.mouse_ids}, ds_scale=1,
    )


def build_reference_model(cfg: O.Config, sd: O.SD, seed: int):
    import_reference()
    from v1t.models.model import Model

    ds = {m: FakeDS(W.make_coordinates(seed, m, cfg.num_neurons[m]), cfg.num_neurons[m]) for m in cfg.mouse_ids}
    model = Model(ref_args(cfg), ds=ds)
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert set(res.missing_keys) <= {"image_cropper.grid", "elu1.one"}, res.missing_keys
    return model


def check(name: str, ref: torch.Tensor, got: torch.Tensor, rtol: float, atol: float):
    ref = ref.detach().to(torch.float64)
    got = got.detach().to(torch.float64)
    err = (ref - got).abs()
    bound = atol + rtol * ref.abs()
    worst = float((err / bound).max())
    if not (worst <= 1.0):
        raise SystemExit(f"ORACLE != REFERENCE at {name}: max err {float(err.max()):.3e} (x{worst:.2f} of bound)")
    return float(err.max())


def oracle_grads(cfg, sd, batch, mouse_id, ds_size, eps=None, dtype=torch.float32):
    sdd = {k: (v.to(dtype).clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    b = {k: v.to(dtype) for k, v in batch.items()}
    loss, reg, y = O.total_loss(cfg, sdd, b, mouse_id, ds_size, eps=None if eps is None else eps.to(dtype))
    (loss + reg).backward()
    keys = O.core_param_keys(sd) + O.readout_param_keys(sd, mouse_id) + O.shifter_param_keys(sd, mouse_id) + O.image_shifter_param_keys(sd, mouse_id)
    # a parameter the restatement never touches (sigma in eval mode) has an all-zero gradient in the reference
    return loss.detach(), reg.detach(), y.detach(), {k: (sdd[k].grad if sdd[k].grad is not None else torch.zeros_like(sdd[k])) for k in keys}


def ref_forward_backward(model, cfg, batch, mouse_id, ds_size, train_eps_seed=None):
    """Reference fwd (+ Poisson loss + regulariser, train.py:59-73) and backward. eval mode unless
    train_eps_seed is given (then train mode with dropout 0 configured by cfg, eps drawn from that seed)."""
    from v1t.losses import PoissonLoss

    crit = PoissonLoss(SimpleNamespace(ds_scale=1), ds={m: SimpleNamespace(dataset=range(int(ds_size))) for m in cfg.mouse_ids})
    model.zero_grad(set_to_none=True)
    model.train(train_eps_seed is not None)
    taps = {}
    hooks = []
    core = model.core
    tokenizer = core.tokenizer if cfg.core == "cct" else core.patch_embedding
    hooks.append(tokenizer.register_forward_hook(lambda m, i, o: taps.__setitem__("patch_embed", o.detach().clone())))
    if train_eps_seed is not None:
        torch.manual_seed(train_eps_seed)
    y, _, _ = model(inputs=batch["image"], mouse_id=mouse_id, behaviors=batch["behavior"], pupil_centers=batch["pupil_center"])
    for h in hooks:
        h.remove()
    loss = crit(y_true=batch["response"], y_pred=y, mouse_id=mouse_id, batch_size=batch["image"].shape[0])
    reg = model.regularizer(mouse_id)
    (loss + reg).backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    return loss.detach(), reg.detach(), y.detach(), grads, taps


def ref_core_taps(model, batch, mouse_id):
    """Core output + per-block residual stream captured by hooks (eval)."""
    model.train(False)
    taps = {}
    hs = []
    for k, blk in enumerate(model.core.transformer.blocks):
        mha, mlp = (blk.mha, blk.mlp) if hasattr(blk, "mha") else (blk["mha"], blk["mlp"])  # cct.py's block is a Module, vit.py's a ModuleDict
        hs.append(mha.register_forward_hook(lambda m, i, o, k=k: taps.__setitem__(f"mha{k}", (o + i[0]).detach().clone())))
        hs.append(mlp.register_forward_hook(lambda m, i, o, k=k: taps.__setitem__(f"mlp{k}", (o + i[0]).detach().clone())))
    with torch.no_grad():
        z = model.core(inputs=batch["image"], mouse_id=mouse_id, behaviors=batch["behavior"], pupil_centers=batch["pupil_center"])
    for h in hs:
        h.remove()
    taps["core"] = z
    return taps


# --------------------------------------------------------------------------------------
# fixtures
# --------------------------------------------------------------------------------------
def gen_train_fixture(name: str, cfg: O.Config, batch_size: int, seed: int, out: dict, train: bool = False, log=print, sd_fn=None):
    """G1/G2/G4: outputs, loss, reg and gradient samples for one mouse-batch."""
    mouse = cfg.mouse_ids[0]
    sd = (sd_fn or W.make_state_dict)(cfg, seed)
    batch = W.make_batch(cfg, mouse, batch_size, seed)
    model = build_reference_model(cfg, sd, seed)
    ds_size = 4500.0
    eps = None
    eps_seed = None
    if train:
        eps_seed = 4321
        torch.manual_seed(eps_seed)
        eps = torch.empty(batch_size, cfg.num_neurons[mouse], 1, 2).normal_().reshape(batch_size, -1, 2)
    loss, reg, y, grads, taps = ref_forward_backward(model, cfg, batch, mouse, ds_size, train_eps_seed=eps_seed)
    # --- pin the oracle
    for dt, rt, at in ((torch.float32, 2e-4, 2e-5), (torch.float64, 2e-5, 2e-6)):
        ol, orr, oy, og = oracle_grads(cfg, sd, batch, mouse, ds_size, eps=eps, dtype=dt)
        e1 = check(f"{name}.y[{dt}]", y, oy, rt, at)
        check(f"{name}.loss[{dt}]", loss, ol, rt, at)
        check(f"{name}.reg[{dt}]", reg, orr, rt, at)
        eg = 0.0
        for k, g in grads.items():
            scale = float(g.abs().max()) + 1e-12
            eg = max(eg, check(f"{name}.grad[{k}][{dt}]", g, og[k], rt * 5, at * 5 + rt * scale) / scale)
        assert set(grads) == set(og), set(grads) ^ set(og)
        log(f"  {name}: oracle[{str(dt)[6:]}] vs reference: y err {e1:.2e}, worst grad err/scale {eg:.2e}")
    if not train:
        rt = ref_core_taps(model, batch, mouse)
        otaps = {}
        with torch.no_grad():
            O.model_forward(cfg, sd, batch["image"], mouse, batch["behavior"], batch["pupil_center"], taps=otaps)
        for k in rt:
            check(f"{name}.tap[{k}]", rt[k], otaps[k], 2e-4, 2e-5)
            out[f"{name}/tap/{k}"] = sample(rt[k])
        check(f"{name}.tap[patch_embed]", taps["patch_embed"], otaps["patch_embed"], 2e-4, 2e-5)
        out[f"{name}/tap/patch_embed"] = sample(taps["patch_embed"])
    else:
        out[f"{name}/eps"] = eps.numpy()
    out[f"{name}/y"] = y.numpy()
    out[f"{name}/loss"] = np.float64(loss.item())
    out[f"{name}/reg"] = np.float64(reg.item())
    for k, g in grads.items():
        out[f"{name}/grad/{k}"] = sample(g)
        out[f"{name}/gradnorm/{k}"] = np.float64(g.double().norm().item())


def gen_variants(out: dict, log=print):
    """G3: outputs only for the constructor variants the reference supports on this path."""
    base = dict(num_blocks=1, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=("A", "B"), num_neurons={"A": 200, "B": 123})
    variants = {
        "beh0": dict(behavior_mode=0),
        "beh2": dict(behavior_mode=2),
        "beh4": dict(behavior_mode=4),
        "franke": dict(input_shape=(2, 36, 64)),
        "nogridpred": dict(disable_grid_predictor=True),
        "grid3": dict(grid_predictor_dim=3),
        "lsa": dict(use_lsa=True),
        "nobias": dict(disable_bias=True),
        "patch1": dict(patch_mode=1),
        "patch2": dict(patch_mode=2),
        "patch3": dict(patch_mode=3),
        "stride2": dict(patch_stride=2),
        "noshift": dict(shift_mode=0),
        "heads3_d40": dict(num_heads=3, emb_dim=40, mlp_dim=72),
    }
    for vn, kw in variants.items():
        cfg = O.Config(**{**base, **kw})
        seed = 77
        sd = W.make_state_dict(cfg, seed)
        model = build_reference_model(cfg, sd, seed)
        model.train(False)
        for mouse in ("A", "B"):
            batch = W.make_batch(cfg, mouse, 2, seed)
            with torch.no_grad():
                y, _, _ = model(inputs=batch["image"], mouse_id=mouse, behaviors=batch["behavior"], pupil_centers=batch["pupil_center"])
                oy = O.model_forward(cfg, sd, batch["image"], mouse, batch["behavior"], batch["pupil_center"])
            e = check(f"variant.{vn}.{mouse}", y, oy, 2e-4, 2e-5)
            out[f"variant/{vn}/{mouse}/y"] = y.numpy()
        log(f"  variant {vn}: ok (last err {e:.2e})")


def gen_rollout(out: dict, log=print):
    """G5: Recorder + attention_rollouts (attention_rollout.py) on a 2-block D=64 model, B=2."""
    import_reference()
    from v1t.utils.attention_rollout import Recorder, attention_rollouts

    cfg = O.Config(num_blocks=2, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=("A",), num_neurons={"A": 64})
    seed = 99
    sd = W.make_state_dict(cfg, seed)
    model = build_reference_model(cfg, sd, seed)
    model.train(False)
    batch = W.make_batch(cfg, "A", 2, seed)
    rec = Recorder(model.core)
    with torch.no_grad():
        _, attn = rec(images=batch["image"], behaviors=batch["behavior"], pupil_centers=batch["pupil_center"], mouse_id="A")
        heat = attention_rollouts(attn, image_shape=list(batch["image"].shape[2:]))
    rec.eject()
    record: t.List[torch.Tensor] = []
    with torch.no_grad():
        O.vit_tokens(cfg, sd, batch["image"], "A", batch["behavior"], batch["pupil_center"], record=record)
    oattn = torch.stack(record, dim=1)
    check("rollout.attn", attn, oattn, 2e-4, 1e-7)
    rows = []
    for i in range(2):
        row = O.attention_rollout_row(oattn[i])
        # reference pre-normalisation row, recomputed from the reference's recorded attention
        a = attn[i].max(dim=1).values + torch.eye(attn.shape[-1])
        a = a / a.sum(-1, keepdim=True)
        j = a[0]
        for n in range(1, a.shape[0]):
            j = a[n] @ j
        check(f"rollout.row{i}", j[0, 1:], row, 1e-4, 1e-9)
        oh = O.attention_rollout(oattn[i], tuple(batch["image"].shape[2:]))
        check(f"rollout.heat{i}", heat[i], oh, 1e-3, 2e-4)
        rows.append(j[0, 1:].numpy())
    out["rollout/row"] = np.stack(rows)
    out["rollout/heatmap"] = heat.numpy()
    out["rollout/attn_sample"] = sample(attn)
    log("  rollout: ok")


def gen_step(out: dict, log=print):
    """G6: one optimizer step = sum of 2 mice (train.py:97-111), AdamW (train.py:216-223), dropout 0,
    eval-style readout (sample=False is not reachable through Model.forward, so eps is drawn from a seed)."""
    cfg = O.Config(num_blocks=1, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=("A", "B"), num_neurons={"A": 200, "B": 123},
                   p_dropout=0.0, t_dropout=0.0)
    seed = 55
    sd = W.make_state_dict(cfg, seed)
    model = build_reference_model(cfg, sd, seed)
    from v1t.losses import PoissonLoss

    ds_size = 4500.0
    crit = PoissonLoss(SimpleNamespace(ds_scale=1), ds={m: SimpleNamespace(dataset=range(int(ds_size))) for m in cfg.mouse_ids})
    lr = 1.647e-3
    opt = torch.optim.AdamW(model.get_parameters(core_lr=lr), lr=lr, betas=(0.9, 0.9999), eps=1e-8, weight_decay=0)
    model.train(True)
    opt.zero_grad()
    eps_all = {}
    for i, mouse in enumerate(cfg.mouse_ids):
        batch = W.make_batch(cfg, mouse, 4, seed)
        torch.manual_seed(1000 + i)
        eps_all[mouse] = torch.empty(4, cfg.num_neurons[mouse], 1, 2).normal_().reshape(4, -1, 2)
        torch.manual_seed(1000 + i)
        y, _, _ = model(inputs=batch["image"], mouse_id=mouse, behaviors=batch["behavior"], pupil_centers=batch["pupil_center"])
        loss = crit(y_true=batch["response"], y_pred=y, mouse_id=mouse, batch_size=4)
        (loss + model.regularizer(mouse)).backward()
    opt.step()
    new = {k: v.detach().clone() for k, v in model.state_dict().items()}
    # oracle step
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    for mouse in cfg.mouse_ids:
        batch = W.make_batch(cfg, mouse, 4, seed)
        l, r, _ = O.total_loss(cfg, osd, batch, mouse, ds_size, eps=eps_all[mouse])
        (l + r).backward()
    pkeys = O.core_param_keys(sd)
    for m in cfg.mouse_ids:
        pkeys += O.readout_param_keys(sd, m) + O.shifter_param_keys(sd, m)
    assert set(pkeys) == {k for k, _ in model.named_parameters()}
    params = {k: osd[k].detach() for k in pkeys}
    O.adamw_step(params, {k: osd[k].grad for k in pkeys}, {}, step=1, lr=lr)
    for k in pkeys:
        check(f"step.{k}", new[k], params[k], 1e-4, 1e-6)
        out[f"step/param/{k}"] = sample(new[k])
    for m in cfg.mouse_ids:
        out[f"step/eps/{m}"] = eps_all[m].numpy()
    log(f"  step: ok ({len(pkeys)} params)")


def gen_resize(out: dict, log=print):
    """Pins the (stub-dependent) 144x256 -> 36x64 bilinear resize of the cropper stage separately."""
    x = torch.from_numpy(np.random.default_rng(5).standard_normal((2, 1, 144, 256)).astype(np.float32))
    ref = F.interpolate(x, size=(36, 64), mode="bilinear", align_corners=False, antialias=False)
    check("resize", ref, O.resize_bilinear(x, (36, 64)), 1e-5, 1e-6)
    out["resize/out_sample"] = sample(ref)
    log("  resize: ok")


def gen_elu_edge(out: dict, log=print):
    """ELU1 + Poisson loss at very negative pre-activations (expm1 quantisation; gradient from the input)."""
    import_reference()
    from v1t.losses import PoissonLoss
    from v1t.models.utils import ELU1

    u = torch.tensor([[-30.0, -20.0, -16.0, -15.0, -5.0, -1e-3, 0.0, 1e-3, 3.0, 15.0]], requires_grad=True)  # (-17 is skipped: expm1 there differs by 1 ulp between ATen code paths and log() amplifies it)
    y = torch.tensor([[0.0, 0.5, 1.0, 2.0, 0.1, 0.0, 1.0, 3.0, 0.2, 9.0]])
    crit = PoissonLoss(SimpleNamespace(ds_scale=1), ds={"A": SimpleNamespace(dataset=range(4500))})
    yh = ELU1()(u)
    loss = crit(y_true=y, y_pred=yh, mouse_id="A", batch_size=16)
    loss.backward()
    uo = u.detach().clone().requires_grad_(True)
    yo = O.elu1(uo)
    lo = O.poisson_loss(y, yo, 4500.0, 16)
    lo.backward()
    # expm1 implementations differ by one fp32 ulp of 1.0 near u = -17 (vectorised vs scalar): absolute floor 1.2e-7
    check("elu_edge.yhat", yh, yo, 1e-6, 1.2e-7)
    check("elu_edge.loss", loss, lo, 1e-5, 0)
    check("elu_edge.grad", u.grad, uo.grad, 1e-6, 1e-12)
    out["elu_edge/u"], out["elu_edge/y_true"] = u.detach().numpy(), y.numpy()
    out["elu_edge/yhat"], out["elu_edge/loss"], out["elu_edge/du"] = yh.detach().numpy(), np.float64(loss.item()), u.grad.numpy()
    log("  elu edge: ok")


def gen_drop_path(out: dict, log=print):
    """G7: stochastic depth (DropPath, models/utils.py:121-141; vit.py:360-361) in train mode, every other dropout 0.
    The reference draws torch.rand((B,1,1)) per branch (block order, mha then mlp) and then the readout's eps; the same
    draws are replayed into the oracle as masks."""
    cfg = O.Config(num_blocks=2, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=("A",), num_neurons={"A": 128},
                   p_dropout=0.0, t_dropout=0.0, drop_path=0.3)
    mouse, B, seed, rng_seed, ds_size = "A", 6, 77, 2468, 4500.0
    sd = W.make_state_dict(cfg, seed)
    batch = W.make_batch(cfg, mouse, B, seed)
    model = build_reference_model(cfg, sd, seed)
    keep = 1.0 - cfg.drop_path
    torch.manual_seed(rng_seed)
    dpm = {}
    for k in range(cfg.num_blocks):
        for br in ("mha", "mlp"):
            dpm[(k, br)] = torch.floor(keep + torch.rand((B, 1, 1), dtype=torch.float32)).reshape(B)
    eps = torch.empty(B, cfg.num_neurons[mouse], 1, 2).normal_().reshape(B, -1, 2)
    marr = torch.stack([torch.stack([dpm[(k, "mha")], dpm[(k, "mlp")]]) for k in range(cfg.num_blocks)])  # (NB, 2, B)
    assert 0 < float(marr.sum()) < marr.numel(), "pick a seed that drops some and keeps some"
    loss, reg, y, grads, _ = ref_forward_backward(model, cfg, batch, mouse, ds_size, train_eps_seed=rng_seed)
    for dt, rt, at in ((torch.float32, 2e-4, 2e-5), (torch.float64, 2e-5, 2e-6)):
        sdd = {k: (v.to(dt).clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
        b = {k: v.to(dt) for k, v in batch.items()}
        ol, orr, oy = O.total_loss(cfg, sdd, b, mouse, ds_size, eps=eps.to(dt), masks={"drop_path": dpm})
        (ol + orr).backward()
        e1 = check(f"g7.y[{dt}]", y, oy, rt, at)
        check(f"g7.loss[{dt}]", loss, ol, rt, at)
        eg = 0.0
        for k, g in grads.items():
            og = sdd[k].grad if sdd[k].grad is not None else torch.zeros_like(sdd[k])
            scale = float(g.abs().max()) + 1e-12
            eg = max(eg, check(f"g7.grad[{k}][{dt}]", g, og, rt * 5, at * 5 + rt * scale) / scale)
        log(f"  g7: oracle[{str(dt)[6:]}] vs reference: y err {e1:.2e}, worst grad err/scale {eg:.2e}")
    out["g7/mask"] = marr.numpy()
    out["g7/eps"] = eps.numpy()
    out["g7/y"] = y.numpy()
    out["g7/loss"] = np.float64(loss.item())
    out["g7/reg"] = np.float64(reg.item())
    for k, g in grads.items():
        out[f"g7/grad/{k}"] = sample(g)
        out[f"g7/gradnorm/{k}"] = np.float64(g.double().norm().item())


def gen_image_shift(out: dict, log=print):
    """G8: center crop < 1 with the learned image shifter (shift_mode 1 / 3 / 4; image_cropper.py:10-47,120-133) from the
    RAW image. Outputs for the three modes, the cropped core input, and every gradient for mode 4 (with shifter /
    cropper L1 so that the image-shifter gradient is the sign term, nearest sampling passing none)."""
    ds_size, B, seed = 4500.0, 3, 91
    for sm in (1, 3, 4):
        cfg = O.Config(num_blocks=1, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=("A", "B"), num_neurons={"A": 96, "B": 50},
                       shift_mode=sm, center_crop=0.8, raw_input_shape=(1, 36, 64), input_shape=(1, int(36 * 0.8), int(64 * 0.8)),
                       shifter_reg_scale=0.01, cropper_reg_scale=0.02)
        name = f"g8/sm{sm}"
        sd = W.make_state_dict(cfg, seed)
        model = build_reference_model(cfg, sd, seed)
        mouse = "B" if sm == 3 else "A"
        batch = W.make_batch(cfg, mouse, B, seed)
        assert tuple(batch["image"].shape[1:]) == (1, 36, 64)
        loss, reg, y, grads, _ = ref_forward_backward(model, cfg, batch, mouse, ds_size)
        model.train(False)
        with torch.no_grad():
            _, img, grid = model(inputs=batch["image"], mouse_id=mouse, behaviors=batch["behavior"], pupil_centers=batch["pupil_center"])
            oshift = O.image_shifter(cfg, sd, mouse, batch["behavior"], batch["pupil_center"])
            oimg = O.crop_nearest(batch["image"], cfg.center_crop, oshift)
        check(f"{name}.shift", grid[:, 0, 0, :] - model.image_cropper.grid[:, 0, 0, :], oshift, 1e-5, 1e-7)
        assert float(oshift.abs().max()) > 0.02, "shifts too small to move the window"
        check(f"{name}.crop", img, oimg, 0.0, 1e-30)  # a gather: bit-exact
        for dt, rt, at in ((torch.float32, 2e-4, 2e-5), (torch.float64, 2e-5, 2e-6)):
            ol, orr, oy, og = oracle_grads(cfg, sd, batch, mouse, ds_size, dtype=dt)
            e1 = check(f"{name}.y[{dt}]", y, oy, rt, at)
            check(f"{name}.loss[{dt}]", loss, ol, rt, at)
            check(f"{name}.reg[{dt}]", reg, orr, rt, at)
            assert set(grads) == set(og), set(grads) ^ set(og)
            eg = 0.0
            for k, g in grads.items():
                scale = float(g.abs().max()) + 1e-12
                eg = max(eg, check(f"{name}.grad[{k}][{dt}]", g, og[k], rt * 5, at * 5 + rt * scale) / scale)
            log(f"  {name}: oracle[{str(dt)[6:]}] vs reference: y err {e1:.2e}, worst grad err/scale {eg:.2e}")
        out[f"{name}/y"] = y.numpy()
        out[f"{name}/shift"] = oshift.numpy()
        out[f"{name}/crop"] = img.numpy()
        out[f"{name}/loss"] = np.float64(loss.item())
        out[f"{name}/reg"] = np.float64(reg.item())
        if sm == 4:
            for k, g in grads.items():
                out[f"{name}/grad/{k}"] = sample(g)
                out[f"{name}/gradnorm/{k}"] = np.float64(g.double().norm().item())
    # center crop alone (shift_mode 2): the fixed window
    cfg = O.Config(num_blocks=1, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=("A",), num_neurons={"A": 96}, center_crop=0.7,
                   raw_input_shape=(1, 36, 64), input_shape=(1, int(36 * 0.7), int(64 * 0.7)))
    sd = W.make_state_dict(cfg, seed)
    model = build_reference_model(cfg, sd, seed)
    batch = W.make_batch(cfg, "A", B, seed)
    model.train(False)
    with torch.no_grad():
        y, img, _ = model(inputs=batch["image"], mouse_id="A", behaviors=batch["behavior"], pupil_centers=batch["pupil_center"])
        oy = O.model_forward_raw(cfg, sd, batch["image"], "A", batch["behavior"], batch["pupil_center"])
    check("g8/crop07.crop", img, O.crop_nearest(batch["image"], 0.7, None), 0.0, 1e-30)
    check("g8/crop07.y", y, oy, 2e-4, 2e-5)
    out["g8/crop07/y"] = y.numpy()
    out["g8/crop07/crop"] = img.numpy()
    log("  g8/crop07: ok")


def gen_metrics(out: dict, log=print):
    """G9: validation / evaluation metrics (train.py:29-39, metrics.py:11-142) of the real reference on a synthetic
    test-tier recording (oracle/metrics_oracle.make_metric_data)."""
    import_reference()
    from oracle import metrics_oracle as MO
    from v1t import losses
    from v1t.metrics import Metrics

    d = MO.make_metric_data()
    yt, yp = torch.from_numpy(d["targets"]), torch.from_numpy(d["predictions"])
    ref = {"metrics/msse": losses.msse(y_true=yt, y_pred=yp), "metrics/poisson_loss": losses.poisson_loss(y_true=yt, y_pred=yp),
           "metrics/single_trial_correlation": torch.mean(losses.correlation(y1=yp, y2=yt, dim=0))}
    mine = MO.compute_metrics(d["targets"], d["predictions"])
    for k, v in ref.items():
        check(f"g9.{k}", v, torch.tensor(mine[k]), 2e-5, 1e-7)
        out[f"g9/{k}"] = np.float64(v.item())
    ds = SimpleNamespace(dataset=SimpleNamespace(tier="test", hashed=False, neuron_ids=d["neuron_ids"].copy()))
    m = Metrics(ds, {"targets": yt, "predictions": yp, "image_ids": torch.from_numpy(d["image_ids"]), "trial_ids": torch.from_numpy(d["trial_ids"])})
    ot, op, oi = MO.order(d["targets"], d["predictions"], d["image_ids"], d["trial_ids"], d["neuron_ids"])
    stc, cta = m.single_trial_correlation(per_neuron=True), m.correlation_to_average(per_neuron=True)
    fev, fe = m._fev(*m.split_responses(), return_exp_var=True)
    check("g9.stc", torch.from_numpy(stc), torch.from_numpy(MO.correlation(op, ot, axis=0)), 1e-5, 1e-6)
    check("g9.cta", torch.from_numpy(cta), torch.from_numpy(MO.correlation_to_average(ot, op, oi)), 1e-5, 1e-6)
    ofev, ofe = MO.fev_feve(ot, op, oi)
    check("g9.fev", torch.from_numpy(fev), torch.from_numpy(ofev), 1e-5, 1e-6)
    check("g9.feve", torch.from_numpy(fe), torch.from_numpy(ofe), 1e-5, 1e-6)
    kept = m.feve(per_neuron=True)
    assert 0.1 < len(kept) / len(fev) < 0.98, "FEV threshold should cut some neurons"
    assert np.array_equal(kept, MO.feve(ot, op, oi)) or np.allclose(kept, MO.feve(ot, op, oi), rtol=1e-5, atol=1e-6)
    out["g9/single_trial_correlation"], out["g9/correlation_to_average"] = stc, cta
    out["g9/fev"], out["g9/feve"], out["g9/feve_kept"] = fev, fe, kept
    log(f"  g9: ok (mean stc {stc.mean():.4f}, cta {cta.mean():.4f}, feve {kept.mean():.4f} over {len(kept)}/{len(fev)} neurons)")


def gen_checkpoint(out: dict, log=print):
    """G11: the checkpoint the reference's Scheduler writes (utils/scheduler.py:84-104) after the G6 step (2 mice, AdamW,
    dropout 0, eps from G6): model keys, optimizer param-group structure and AdamW moments, scheduler state keys."""
    import tempfile

    import_reference()
    from v1t.losses import PoissonLoss
    from v1t.utils.scheduler import Scheduler

    cfg = O.Config(num_blocks=1, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=("A", "B"), num_neurons={"A": 200, "B": 123},
                   p_dropout=0.0, t_dropout=0.0)
    seed, ds_size, lr = 55, 4500.0, 1.647e-3
    sd = W.make_state_dict(cfg, seed)
    model = build_reference_model(cfg, sd, seed)
    crit = PoissonLoss(SimpleNamespace(ds_scale=1), ds={m: SimpleNamespace(dataset=range(int(ds_size))) for m in cfg.mouse_ids})
    opt = torch.optim.AdamW(model.get_parameters(core_lr=lr), lr=lr, betas=(0.9, 0.9999), eps=1e-8, weight_decay=0)
    model.train(True)
    opt.zero_grad()
    for i, mouse in enumerate(cfg.mouse_ids):
        batch = W.make_batch(cfg, mouse, 4, seed)
        torch.manual_seed(1000 + i)  # the same draws as G6 (gen_step)
        y, _, _ = model(inputs=batch["image"], mouse_id=mouse, behaviors=batch["behavior"], pupil_centers=batch["pupil_center"])
        (crit(y_true=batch["response"], y_pred=y, mouse_id=mouse, batch_size=4) + model.regularizer(mouse)).backward()
    opt.step()
    with tempfile.TemporaryDirectory() as d:
        sch = Scheduler(SimpleNamespace(output_dir=d, device=torch.device("cpu"), verbose=0), model=model, optimizer=opt, mode="max")
        assert sch.step(0.25, epoch=3) is False  # better than -inf: writes the checkpoint
        ck = torch.load(os.path.join(d, "ckpt", "model_state.pt"), weights_only=False)
    assert set(ck) == {"epoch", "value", "model", "optimizer", "scheduler"}
    out["g11/epoch"], out["g11/value"] = np.int64(ck["epoch"]), np.float64(ck["value"])
    out["g11/model_keys"] = np.array(list(ck["model"].keys()))
    out["g11/scheduler_keys"] = np.array(sorted(ck["scheduler"].keys()))
    names = [k for k, _ in model.named_parameters()]
    pid = {id(p): k for k, p in model.named_parameters()}
    order = [pid[id(p)] for g in opt.param_groups for p in g["params"]]  # optimizer index -> parameter name
    out["g11/opt_param_names"] = np.array(order)
    out["g11/group_names"] = np.array([g["name"] for g in ck["optimizer"]["param_groups"]])
    out["g11/group_sizes"] = np.array([len(g["params"]) for g in ck["optimizer"]["param_groups"]])
    out["g11/group_lr"] = np.array([g["lr"] for g in ck["optimizer"]["param_groups"]])
    for i, k in enumerate(order):
        st = ck["optimizer"]["state"][i]
        out[f"g11/exp_avg/{k}"] = sample(st["exp_avg"])
        out[f"g11/exp_avg_sq/{k}"] = sample(st["exp_avg_sq"])
        assert float(st["step"]) == 1.0
    assert sorted(order) == sorted(names)
    log(f"  g11: ok ({len(order)} parameters in {len(ck['optimizer']['param_groups'])} groups)")


def gen_boundary(out: dict, log=print):
    """G12: what the reference's own `Model` / `Gaussian2DReadout` constructors produce at the drop-in boundary
    (model.py:74-139, gaussian2d.py:19-81,138-186): optimizer group names and the parameter order inside them for
    shift_mode 2 and 4 (torch.optim state dicts are positional), and the default initialisation of the readout
    (sigma ~ U(+-0.1), features = 1/C, bias by bias_mode, mu predictor / free mu) under a fixed torch seed."""
    import_reference()
    from v1t.models.model import Model
    from v1t.models.readout.gaussian2d import Gaussian2DReadout

    for sm, extra in ((2, {}), (4, dict(center_crop=0.8, raw_input_shape=(1, 36, 64), input_shape=(1, 28, 51)))):
        cfg = O.Config(num_blocks=1, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=("A", "B"), num_neurons={"A": 40, "B": 23}, shift_mode=sm, **extra)
        ds = {m: FakeDS(W.make_coordinates(3, m, cfg.num_neurons[m]), cfg.num_neurons[m]) for m in cfg.mouse_ids}
        model = Model(ref_args(cfg), ds=ds)
        pid = {id(p): k for k, p in model.named_parameters()}
        groups = model.get_parameters(core_lr=1e-3)
        out[f"g12/sm{sm}/group_names"] = np.array([g["name"] for g in groups])
        for g in groups:
            out[f"g12/sm{sm}/group/{g['name']}"] = np.array([pid[id(p)] for p in g["params"]])
        out[f"g12/sm{sm}/core_output_shape"] = np.asarray(model.core.output_shape)
        out[f"g12/sm{sm}/state_keys"] = np.array(list(model.state_dict().keys()))
    n, c = 57, 24
    coords = W.make_coordinates(5, "A", n)
    stats = {"mean": (0.5 + np.arange(n, dtype=np.float32) / n), "std": (1.0 + 0.5 * np.cos(np.arange(n, dtype=np.float32)) ** 2).astype(np.float32)}
    for tag, kw in (("bias0", dict(bias_mode=0)), ("bias1", dict(bias_mode=1)), ("bias2", dict(bias_mode=2)), ("freemu", dict(bias_mode=0, disable_grid_predictor=True)),
                    ("grid3", dict(bias_mode=0, grid_predictor_dim=3))):
        a = SimpleNamespace(readout_reg_scale=0.0076, disable_grid_predictor=False, grid_predictor_dim=2, bias_mode=0)
        for k, v in kw.items():
            setattr(a, k, v)
        ds = SimpleNamespace(dataset=SimpleNamespace(coordinates=coords, response_stats=stats))
        torch.manual_seed(77)
        ro = Gaussian2DReadout(a, input_shape=(c, 5, 7), output_shape=(n,), ds=ds, name="x")
        for k, v in ro.state_dict().items():
            out[f"g12/init/{tag}/{k}"] = v.detach().numpy().copy()
    log("  g12: ok")


def gen_data(out: dict, log=print):
    """G10: the reference's MiceDataset + DataLoader (data.py:275-491) over a tiny recording written in the on-disk layout by
    oracle/fake_sensorium.py (same seeds in the tests): collated, standardised batches per tier."""
    import tempfile

    import_reference()
    from torch.utils.data import DataLoader
    from v1t import data as RD

    from oracle import fake_sensorium as FS

    with tempfile.TemporaryDirectory() as root:
        for ds_name, mouse, shape, gray in (("sensorium", "A", (1, 12, 16), False), ("franke2022", "F", (2, 6, 8), True)):
            FS.write_fake_mouse(root, ds_name, mouse, seed=3, trials=23, image_shape=shape, neurons=9)
            args = SimpleNamespace(ds_name=ds_name, behavior_mode=3, seed=1, gray_scale=gray, verbose=0, limit_data=None)
            for tier in ("train", "validation", "test"):
                dsx = RD.MiceDataset(args, tier=tier, data_dir=root, mouse_id=mouse)
                batches = list(DataLoader(dsx, batch_size=4, shuffle=False))
                tag = f"g10/{ds_name}/{tier}"
                out[f"{tag}/n"] = np.int64(len(dsx))
                for k in ("image", "response", "behavior", "pupil_center", "image_id", "trial_id"):
                    out[f"{tag}/{k}"] = torch.cat([b[k] for b in batches]).numpy()
                assert out[f"{tag}/image"].dtype == np.float32 and out[f"{tag}/response"].dtype == np.float32
                out[f"{tag}/precision"] = np.asarray(dsx._response_precision)
                out[f"{tag}/image_shape"] = np.asarray(dsx.image_shape)
            log(f"  g10/{ds_name}: ok ({len(dsx)} test trials, image {dsx.image_shape})")


def input_gradient(name: str, cfg: O.Config, sd, batch_size: int, seed: int, out: dict, ds_size: float = 4500.0, log=print):
    """d (loss + reg) / d image from the REAL reference's autograd (the core is plain torch there: Unfold + Linear vit.py:66-72, Conv2d
    :73-82, patch LayerNorms :83-100; conv tokenizer cct.py:30-104; the cropper at crop 1 is the identity gather), eval mode, whole tensor
    stored; the oracle's own autograd is pinned against it (fp32 and fp64). SURVEY 8(c) G1 "grads of all params + core input"."""
    from v1t.losses import PoissonLoss

    mouse = cfg.mouse_ids[0]
    batch = W.make_batch(cfg, mouse, batch_size, seed)
    model = build_reference_model(cfg, sd, seed)
    model.train(False)
    crit = PoissonLoss(SimpleNamespace(ds_scale=1), ds={m: SimpleNamespace(dataset=range(int(ds_size))) for m in cfg.mouse_ids})
    img = batch["image"].clone().requires_grad_(True)
    y, _, _ = model(inputs=img, mouse_id=mouse, behaviors=batch["behavior"], pupil_centers=batch["pupil_center"])
    loss = crit(y_true=batch["response"], y_pred=y, mouse_id=mouse, batch_size=batch_size)
    (loss + model.regularizer(mouse)).backward()
    dx = img.grad.detach().clone()
    for dt, rt in ((torch.float32, 1e-3), (torch.float64, 1e-4)):
        b = {k: v.to(dt) for k, v in batch.items()}
        b["image"] = b["image"].clone().requires_grad_(True)
        ol, orr, _ = O.total_loss(cfg, O.to_dtype(sd, dt), b, mouse, ds_size)
        (ol + orr).backward()
        e = check(f"{name}.dx[{dt}]", dx, b["image"].grad, rt, rt * float(dx.abs().max()))
    out[f"{name}/input_grad"] = dx.numpy()
    log(f"  {name}: d / d image |max| {float(dx.abs().max()):.3e}, oracle[fp64] vs reference {e:.2e}")


def gen_sharp_and_input_grad(out: dict, log=print):
    """G14: (i) the default V1T in the regime of TRAINED weights (`weights.make_sharp_state_dict`: score std 3-9, LayerNorm gains 0.3-3 with
    sign flips, residual outlier channels of +-80, clamped / out-of-range sample positions) - predictions, loss, every gradient, eval and
    train-mode sampling with injected eps; (ii) gradients with respect to the core input for C1, the default V1T (flat and sharp), the
    patch modes, a stride, two channels and the CCT tokenizer."""
    cfg = W.config_c2({"A": 8000})
    gen_train_fixture("g14", cfg, 2, 1234, out, log=log, sd_fn=W.make_sharp_state_dict)
    c = W.config_c2({"A": 8000})
    c.p_dropout = 0.0
    c.t_dropout = 0.0
    gen_train_fixture("g14t", c, 2, 1234, out, train=True, log=log, sd_fn=W.make_sharp_state_dict)
    input_gradient("g14", cfg, W.make_sharp_state_dict(cfg, 1234), 2, 1234, out, log=log)
    input_gradient("g1", W.config_c1(), W.make_state_dict(W.config_c1(), 1234), 2, 1234, out, log=log)
    input_gradient("g2", cfg, W.make_state_dict(cfg, 1234), 2, 1234, out, log=log)
    base = dict(num_blocks=1, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=("A", "B"), num_neurons={"A": 200, "B": 123})
    for vn, kw in {"patch1": dict(patch_mode=1), "patch2": dict(patch_mode=2), "patch3": dict(patch_mode=3), "stride2": dict(patch_stride=2),
                   "franke": dict(input_shape=(2, 36, 64))}.items():
        c = O.Config(**{**base, **kw})
        input_gradient(f"dx_{vn}", c, W.make_state_dict(c, 77), 2, 77, out, log=log)
    c13 = W.config_cct({"A": 200})
    c13.num_blocks, c13.behavior_mode, c13.pos_emb, c13.emb_dim, c13.mlp_dim, c13.num_heads = 2, 0, "none", 64, 128, 2
    input_gradient("dx_cct", c13, W.make_state_dict(c13, 77), 2, 77, out, log=log)


def main():
    torch.set_num_threads(8)
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    torch.manual_seed(0)

    def save(fname, d):
        path = os.path.join(ROOT, "tests", "golden", fname)
        np.savez_compressed(path, **d)
        print(f"wrote {path}: {os.path.getsize(path) / 1e3:.1f} kB, {len(d)} arrays")

    d = {}
    print("G14 trained-regime (sharp attention, outlier channels, clamped positions) default V1T + gradients w.r.t. the core input")
    gen_sharp_and_input_grad(d)
    save("g14_sharp_dx.npz", d)
    if "--only-g14" in sys.argv:
        return

    d = {}
    print("G12 boundary: optimizer groups / readout initialisation of the reference's constructors")
    gen_boundary(d)
    save("g12_boundary.npz", d)
    if "--only-g12" in sys.argv:
        return

    d = {}
    print("G11 checkpoint written by the reference's Scheduler")
    gen_checkpoint(d)
    save("g11_checkpoint.npz", d)
    if "--only-g11" in sys.argv:
        return

    d = {}
    print("G10 data path (MiceDataset + DataLoader over the on-disk layout)")
    gen_data(d)
    save("g10_data.npz", d)
    if "--only-g10" in sys.argv:
        return

    d = {}
    print("G9 validation / evaluation metrics")
    gen_metrics(d)
    save("g9_metrics.npz", d)
    if "--only-g9" in sys.argv:
        return

    d = {}
    print("G8 center crop + learned image shifter (shift_mode 1/3/4)")
    gen_image_shift(d)
    save("g8_image_shift.npz", d)
    if "--only-g8" in sys.argv:
        return

    d = {}
    print("G13 CCT core (core/cct.py) at the reference's default CCT arguments, mouse A x 500 neurons, B=2, eval; + a 2-block variant without behaviour")
    gen_train_fixture("g13", W.config_cct(), 2, 1234, d)
    c13 = W.config_cct({"A": 200})
    c13.num_blocks, c13.behavior_mode, c13.pos_emb, c13.emb_dim, c13.mlp_dim, c13.num_heads = 2, 0, "none", 64, 128, 2
    gen_train_fixture("g13b", c13, 2, 77, d)
    c13 = W.config_cct({"A": 200, "B": 123})
    c13.num_blocks, c13.behavior_mode, c13.emb_dim, c13.mlp_dim, c13.mouse_ids, c13.input_shape = 1, 4, 144, 96, ("A", "B"), (2, 36, 64)
    gen_train_fixture("g13c", c13, 2, 78, d)
    save("g13_cct.npz", d)
    if "--only-g13" in sys.argv:
        return

    d = {}
    print("G7 drop_path (train mode, stochastic depth 0.3)")
    gen_drop_path(d)
    save("g7_drop_path.npz", d)
    if "--only-g7" in sys.argv:
        return

    d = {}
    print("G1 (C1: 1 block, D=64, 256 neurons, B=2, eval)")
    gen_train_fixture("g1", W.config_c1(), 2, 1234, d)
    print("G4 (C1, train-mode readout sampling with injected eps, dropout 0)")
    c = W.config_c1()
    c.p_dropout = 0.0
    c.t_dropout = 0.0
    gen_train_fixture("g4", c, 2, 1234, d, train=True)
    save("g1_g4_c1.npz", d)

    d = {}
    print("G2 (C2 default V1T, mouse A x 8000 neurons, B=2, eval)")
    gen_train_fixture("g2", W.config_c2({"A": 8000}), 2, 1234, d)
    print("G2b (C4 Franke-shaped, 2 channels, 1121 neurons, B=2, eval)")
    gen_train_fixture("g2b", W.config_c4(), 2, 1234, d)
    save("g2_default.npz", d)

    d = {}
    print("G3 variants")
    gen_variants(d)
    print("G5 rollout")
    gen_rollout(d)
    print("G6 optimizer step")
    gen_step(d)
    gen_resize(d)
    gen_elu_edge(d)
    save("g3_g5_g6.npz", d)


if __name__ == "__main__":
    main()
