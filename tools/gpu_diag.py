"""GPU bring-up diagnostics: runs every kernel against a torch/oracle reference and PRINTS the errors
(does not stop at the first failure). `python tools/gpu_diag.py [--quick]` -> stdout; used through gpurun.
Test infrastructure (imports the oracle as the checker)."""
from __future__ import annotations

import math
import os
import sys
import time
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import v1t_amd  # noqa: E402
from v1t_amd import lib as L  # noqa: E402
from oracle import v1t_oracle as O  # noqa: E402
from oracle import weights as W  # noqa: E402

dev = torch.device("cuda:0")
lib = L.load()


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30)), float((a - b).abs().max())


def report(name, a, b):
    r, e = rel(a, b)
    bad = not math.isfinite(r) or r > 3e-2
    print(f"{'FAIL' if bad else 'ok  '} {name:44s} rel {r:.3e} abs {e:.3e} (ref max {float(b.abs().max()):.3e})", flush=True)
    return r


def section(fn):
    print(f"\n=== {fn.__name__}", flush=True)
    try:
        t0 = time.time()
        fn()
        torch.cuda.synchronize()
        print(f"    ({time.time() - t0:.1f}s)", flush=True)
    except Exception:
        traceback.print_exc()
        sys.stdout.flush()


def gemm_nt():
    g = torch.Generator().manual_seed(1)
    for (M, N, K) in ((300, 160, 160), (1000, 1920, 160), (257, 512, 160), (700, 160, 640), (129, 64, 256), (64, 768, 64)):
        A = torch.randn(M, K, generator=g).to(dev).bfloat16()
        B = torch.randn(N, K, generator=g).to(dev).bfloat16()
        ref = A.float() @ B.float().t()
        C = torch.empty(M, N, device=dev, dtype=torch.float32)
        L.check(lib.v1t_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, M, N, K, C.data_ptr(), N, 1, L.stream()))
        report(f"gemm_nt f32 {M}x{N}x{K}", C, ref)
        Cb = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        L.check(lib.v1t_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, M, N, K, Cb.data_ptr(), N, 0, L.stream()))
        report(f"gemm_nt bf16 {M}x{N}x{K}", Cb.float(), ref)


def gemm_tn():
    g = torch.Generator().manual_seed(2)
    for (M, NY, NX, mc) in ((1000, 160, 512, 256), (3000, 1920, 160, 1024), (500, 512, 160, 128), (777, 64, 256, 128), (100, 160, 640, 128)):
        Y = torch.randn(M, NY, generator=g).to(dev).bfloat16()
        X = torch.randn(M, NX, generator=g).to(dev).bfloat16()
        ref = Y.float().t() @ X.float()
        dW = torch.zeros(NY, NX, device=dev)
        L.check(lib.v1t_gemm_tn(Y.data_ptr(), NY, X.data_ptr(), NX, M, NY, NX, dW.data_ptr(), NX, mc, L.stream()))
        report(f"gemm_tn {M} {NY}x{NX}", dW, ref)


def attn_ref(qkv, B, H, T, DP, scale, mask=None, p=0.0, diag=False):
    q, k, v = qkv.float().view(B, T, 3, H, DP).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * scale
    if diag:
        s = s.masked_fill(torch.eye(T, dtype=torch.bool, device=s.device), -torch.finfo(torch.float32).max)
    a = torch.softmax(s, -1)
    if mask is not None:
        a = a * mask.view(B, H, T, T).float() / (1 - p)
    return (a @ v).permute(0, 2, 1, 3).reshape(B * T, H * DP)


def attention():
    g = torch.Generator().manual_seed(3)
    for (B, H, T, DP, p) in ((2, 4, 1654, 160, 0.0), (2, 4, 1654, 64, 0.0), (1, 2, 100, 160, 0.0), (2, 3, 333, 64, 0.25), (1, 4, 1654, 160, 0.2544)):
        qkv = (torch.randn(B * T, 3 * H * DP, generator=g) * 0.7).to(dev).bfloat16().requires_grad_(True)
        scale = torch.tensor([DP ** -0.5], device=dev)
        o = torch.empty(B * T, H * DP, device=dev, dtype=torch.bfloat16)
        lse = torch.empty(B, H, T, device=dev)
        seed, sid = 777, 8
        L.check(lib.v1t_attention_forward(qkv.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, seed, sid, o.data_ptr(), lse.data_ptr(), L.stream()))
        mask = None
        if p > 0:
            mask = torch.empty(B * H * T, T, device=dev, dtype=torch.uint8)
            L.check(lib.v1t_dropout_mask(seed, sid, p, B * H * T, T, mask.data_ptr(), L.stream()))
            print(f"     keep fraction {float(mask.float().mean()):.4f} (expect {1 - p:.4f})")
        ref = attn_ref(qkv, B, H, T, DP, float(scale), mask, float(lib.v1t_attention_dropout_rate(p)))
        tag = f"B{B} H{H} T{T} DP{DP} p{p}"
        report(f"attn fwd {tag}", o.float(), ref)
        dO = (torch.randn(B * T, H * DP, generator=g) * 0.5).to(dev).bfloat16()
        (gq,) = torch.autograd.grad(ref, qkv, dO.float())
        dqkv = torch.empty_like(qkv)
        delta = torch.empty(B, H, T, device=dev)
        L.check(lib.v1t_attention_backward(qkv.data_ptr(), o.data_ptr(), dO.data_ptr(), lse.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, seed,
                                           sid, delta.data_ptr(), dqkv.data_ptr(), None, L.stream()))
        gq = gq.view(B * T, 3, H * DP)
        d = dqkv.float().view(B * T, 3, H * DP)
        report(f"attn dq  {tag}", d[:, 0], gq[:, 0])
        report(f"attn dk  {tag}", d[:, 1], gq[:, 1])
        report(f"attn dv  {tag}", d[:, 2], gq[:, 2])


def readout():
    g = torch.Generator().manual_seed(4)
    for (B, C, H, Wd, N) in ((3, 155, 29, 57, 1000), (2, 64, 29, 57, 257)):
        z = torch.randn(B, C, H, Wd, generator=g).to(dev)
        grid = (torch.rand(B, N, 2, generator=g) * 2.4 - 1.2).to(dev)
        FS = (C + 31) // 32 * 32
        feat = torch.zeros(N, FS, device=dev)
        feat[:, :C] = torch.randn(N, C, generator=g).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        zr, gr, fr = z.clone().requires_grad_(True), grid.clone().requires_grad_(True), feat[:, :C].clone().requires_grad_(True)
        ref = (O.bilinear_sample(zr, gr) * fr.t()[None]).sum(1) + bias
        zl = z.permute(0, 2, 3, 1).contiguous()
        out = torch.empty(B, N, device=dev)
        L.check(lib.v1t_gaussian2d_forward(zl.data_ptr(), H * Wd * C, C, B, C, H, Wd, N, grid.data_ptr(), feat.data_ptr(), FS, bias.data_ptr(), out.data_ptr(), L.stream()))
        report(f"readout fwd C{C} N{N}", out, ref)
        go = torch.randn(B, N, generator=g).to(dev)
        gz, gg, gf = torch.autograd.grad(ref, (zr, gr, fr), go)
        dz = torch.zeros_like(zl)
        dgrid = torch.empty_like(grid)
        dfeat = torch.zeros_like(feat)
        dbias = torch.zeros_like(bias)
        L.check(lib.v1t_gaussian2d_backward(zl.data_ptr(), H * Wd * C, C, B, C, H, Wd, N, grid.data_ptr(), feat.data_ptr(), FS, go.data_ptr(), dz.data_ptr(),
                                            H * Wd * C, C, dgrid.data_ptr(), dfeat.data_ptr(), dbias.data_ptr(), L.stream()))
        report("readout dz", dz.permute(0, 3, 1, 2), gz)
        report("readout dgrid", dgrid, gg)
        report("readout dfeat", dfeat[:, :C], gf)
        report("readout dbias", dbias, go.sum(0))


def layernorm():
    g = torch.Generator().manual_seed(5)
    for (B, T, D, DP) in ((2, 1654, 64, 64), (2, 1654, 155, 160), (3, 100, 40, 64)):
        x = torch.zeros(B, T, DP)
        x[:, :, :D] = torch.randn(B, T, D, generator=g)
        inj = torch.zeros(B, DP)
        inj[:, :D] = torch.randn(B, D, generator=g)
        gamma, beta = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
        xd, injd, gd, bd = x.to(dev), inj.to(dev), gamma.to(dev), beta.to(dev)
        xout = torch.empty_like(xd)
        z = torch.empty(B, T, DP, device=dev, dtype=torch.bfloat16)
        mean, rstd = torch.empty(B * T, device=dev), torch.empty(B * T, device=dev)
        L.check(lib.v1t_layernorm_forward(xd.data_ptr(), injd.data_ptr(), xout.data_ptr(), gd.data_ptr(), bd.data_ptr(), z.data_ptr(), mean.data_ptr(),
                                          rstd.data_ptr(), B, T, D, DP, 1e-5, L.stream()))
        xr = (x[:, :, :D] + inj[:, None, :D]).to(dev).requires_grad_(True)
        ref = O.layer_norm(xr, gd.clone().requires_grad_(True), bd)
        report(f"ln fwd D{D}", z[:, :, :D].float(), ref.detach())
        report(f"ln xout D{D}", xout[:, :, :D], xr.detach())
        torch.cuda.synchronize()
        print("     fwd done", flush=True)
        dz = torch.zeros(B, T, DP)
        dz[:, :, :D] = torch.randn(B, T, D, generator=g)
        gin = torch.zeros(B, T, DP)
        gin[:, :, :D] = torch.randn(B, T, D, generator=g)
        dzd, gind = dz.to(dev), gin.to(dev)
        gout = torch.empty_like(gind)
        dgamma, dbeta, dinj = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(B, DP, device=dev)
        dyn = torch.empty(B, T, DP, device=dev, dtype=torch.bfloat16)
        dbn = torch.zeros(D, device=dev)
        L.check(lib.v1t_layernorm_backward(dzd.data_ptr(), xout.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gd.data_ptr(), gind.data_ptr(), gout.data_ptr(),
                                           dgamma.data_ptr(), dbeta.data_ptr(), dinj.data_ptr(), dyn.data_ptr(), dbn.data_ptr(), B, T, D, DP, L.stream()))
        torch.cuda.synchronize()
        gam = gd.clone().requires_grad_(True)
        ref = O.layer_norm(xr, gam, bd)
        gx, gg = torch.autograd.grad(ref, (xr, gam), dzd[:, :, :D])
        report(f"ln bwd gout D{D}", gout[:, :, :D], gx + gind[:, :, :D])
        report(f"ln bwd dgamma D{D}", dgamma, gg)
        report(f"ln bwd dbeta D{D}", dbeta, dzd[:, :, :D].sum((0, 1)))
        report(f"ln bwd dinject D{D}", dinj[:, :D], (gx + gind[:, :, :D]).sum(1))
        report(f"ln bwd dy_next D{D}", dyn[:, :, :D].float(), gx + gind[:, :, :D])
        report(f"ln bwd dbias_next D{D}", dbn, (gx + gind[:, :, :D]).sum((0, 1)))


def build_model(cfg: O.Config, seed=1234, train_dropout=False):
    from v1t_amd.synthetic import default_args, make_ds

    args = default_args(
        input_shape=cfg.input_shape, resize_image=0, num_blocks=cfg.num_blocks, emb_dim=cfg.emb_dim, mlp_dim=cfg.mlp_dim,
        num_heads=cfg.num_heads, behavior_mode=cfg.behavior_mode, use_lsa=cfg.use_lsa, disable_bias=cfg.disable_bias,
        patch_mode=cfg.patch_mode, patch_stride=cfg.patch_stride, shift_mode=cfg.shift_mode,
        disable_grid_predictor=cfg.disable_grid_predictor, grid_predictor_dim=cfg.grid_predictor_dim,
        p_dropout=cfg.p_dropout, t_dropout=cfg.t_dropout,
    )
    args.output_shapes = {m: (cfg.num_neurons[m],) for m in cfg.mouse_ids}
    ds = make_ds(cfg.num_neurons)
    model = v1t_amd.Model(args, ds)
    sd = W.make_state_dict(cfg, seed)
    r = model.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys and set(r.missing_keys) <= {"image_cropper.grid", "elu1.one"}, r
    return model.to(dev), sd


def model_check(cfg: O.Config, name: str, B=2, grads=True, taps=True):
    model, sd = build_model(cfg)
    mouse = cfg.mouse_ids[0]
    batch = W.make_batch(cfg, mouse, B)
    bd = {k: v.to(dev) for k, v in batch.items()}
    model.train(False)
    with torch.no_grad():
        y0, _, _ = model(inputs=bd["image"], mouse_id=mouse, behaviors=bd["behavior"], pupil_centers=bd["pupil_center"])
    y, _, _ = model(inputs=bd["image"], mouse_id=mouse, behaviors=bd["behavior"], pupil_centers=bd["pupil_center"])  # grad enabled: keeps per-block activations
    print("     no_grad vs grad forward identical:", bool(torch.equal(y0, y)))
    y = y.detach()
    otaps = {}
    sdd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    oy = O.model_forward(cfg, sdd, batch["image"], mouse, batch["behavior"], batch["pupil_center"], taps=otaps)
    report(f"{name} y", y, oy.detach())
    core = model.core
    T, DP, D = core.num_tokens, core.padded_dim, cfg.emb_dim
    if taps:
        R = B * T
        x0 = core.workspace_tensor("x0")[: R * DP * 4].view(torch.float32).view(B, T, DP)
        report(f"{name} tap patch_embed", x0[:, :, :D], otaps["patch_embed"].detach())
        print(f"     pad cols max {float(x0[:, :, D:].abs().max()) if DP > D else 0.0:.3e}")
        for k in range(cfg.num_blocks):
            xm = core.workspace_tensor("xm", k)[: R * DP * 4].view(torch.float32).view(B, T, DP)
            report(f"{name} tap mha{k}", xm[:, :, :D], otaps[f"mha{k}"].detach())
    if not grads:
        return
    loss, reg, oyy = O.total_loss(cfg, sdd, batch, mouse, 4500.0)
    (loss + reg).backward()
    from v1t_amd.losses import elu1_poisson_loss

    model.zero_grad(set_to_none=True)
    u, _, _ = model(inputs=bd["image"], mouse_id=mouse, behaviors=bd["behavior"], pupil_centers=bd["pupil_center"], activate=False)
    l, yh = elu1_poisson_loss(u, bd["response"], 4500.0, B)
    total = l + model.regularizer(mouse)
    total.backward()
    report(f"{name} loss", l.detach(), loss.detach())
    worst = 0.0
    for k, p in model.named_parameters():
        if p.grad is None:
            print(f"     no grad: {k}")
            continue
        ref = sdd[k].grad
        if ref is None:
            ref = torch.zeros_like(sdd[k])
        r, e = rel(p.grad, ref)
        worst = max(worst, r)
        flag = "FAIL" if (not math.isfinite(r) or r > 5e-2) else "ok  "
        print(f"{flag} grad {k:60s} rel {r:.3e} abs {e:.3e} refmax {float(ref.abs().max()):.3e}")
    print(f"     worst grad rel err {worst:.3e}")


def model_c1():
    model_check(W.config_c1(), "C1")


def model_c2():
    model_check(W.config_c2({"A": 8000}), "C2")


def model_c4():
    model_check(W.config_c4(), "C4")


def train_dropout_stats():
    cfg = W.config_c1()
    model, sd = build_model(cfg)
    mouse = "A"
    batch = {k: v.to(dev) for k, v in W.make_batch(cfg, mouse, 4).items()}
    model.train(True)
    ys = []
    for _ in range(3):
        y, _, _ = model(inputs=batch["image"], mouse_id=mouse, behaviors=batch["behavior"], pupil_centers=batch["pupil_center"])
        ys.append(y)
    model.train(False)
    ye, _, _ = model(inputs=batch["image"], mouse_id=mouse, behaviors=batch["behavior"], pupil_centers=batch["pupil_center"])
    print("     train-vs-eval mean abs diff", float((ys[0] - ye).abs().mean()), "train-train diff", float((ys[0] - ys[1]).abs().mean()), "finite", bool(torch.isfinite(ys[0]).all()))


def timing():
    from v1t_amd.synthetic import sensorium_config, make_batch
    from v1t_amd.trainer import Trainer

    args, ds = sensorium_config(input_shape=(1, 36, 64), resize_image=0)
    model = v1t_amd.Model(args, ds).to(dev)
    tr = Trainer(args, model, ds)
    batches = {m: make_batch(args, m, 8000, 16, dev, seed=i) for i, m in enumerate(args.mouse_ids)}
    for _ in range(2):
        out = tr.train_step(batches)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 5
    for _ in range(n):
        out = tr.train_step(batches)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    print(f"     train step {dt * 1e3:.2f} ms -> {112 / dt:.1f} images/s; loss {float(out['loss']):.4e}")


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    which = sys.argv[1:] or ["gemm_nt", "gemm_tn", "attention", "readout", "model_c1", "model_c4", "model_c2", "train_dropout_stats", "timing"]
    for w in which:
        section(globals()[w])
