"""GPU: each HIP kernel family through the C-ABI against a plain fp32 torch reference of the same op
(the reference gets the SAME bf16-rounded inputs, so the tolerance covers accumulation order and the
bf16 rounding of internal operands / outputs only)."""
import math

import numpy as np
import pytest
import torch

from tests.helpers import check_grad, check_rel, rel_to_max

pytestmark = pytest.mark.gpu


def exp_env(**switches):
    """Environment of a child process that loads the EXPERIMENT build (libv1t_amd_exp.so = the product's sources + -DV1T_EXPERIMENTS,
    v1t_amd/build.py) with development switches set: the product library contains neither the switches nor the experiment kernels, so every
    "fused == unfused" equality test compares the product (no switch, default library) against an experiment-build variant."""
    import os

    from v1t_amd.build import build_experiments

    build_experiments(force=False, verbose=False)
    env = {k: v for k, v in os.environ.items() if not k.startswith("V1T_")}
    env["V1T_LIB"] = "libv1t_amd_exp.so"
    env.update({k: str(v) for k, v in switches.items()})
    return env


def product_env():
    import os

    return {k: v for k, v in os.environ.items() if not k.startswith("V1T_")}


@pytest.fixture(scope="module")
def ctx():
    from v1t_amd import lib as L

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return L.load(), L, torch.device("cuda:0")


@pytest.mark.parametrize("M,N,K", [(300, 160, 160), (1000, 1920, 160), (257, 512, 160), (700, 160, 640), (129, 64, 256), (64, 768, 64), (1, 32, 32)])
def test_gemm_nt(ctx, M, N, K):
    lib, L, dev = ctx
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dev).bfloat16()
    B = torch.randn(N, K, generator=g).to(dev).bfloat16()
    ref = A.float() @ B.float().t()
    C = torch.empty(M, N, device=dev)
    L.check(lib.v1t_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, M, N, K, C.data_ptr(), N, 1, L.stream()))
    check_rel("test_gemm_nt:0", C.cpu(), ref.cpu(), 2e-6)  # fp32 accumulate of exact bf16 products
    Cb = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    L.check(lib.v1t_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, M, N, K, Cb.data_ptr(), N, 0, L.stream()))
    check_rel("test_gemm_nt:1", Cb.float().cpu(), ref.cpu(), 4e-3)  # one bf16 rounding of the output


@pytest.mark.parametrize("M,NY,NX,mc", [(1000, 160, 512, 256), (3000, 1920, 160, 1024), (500, 512, 160, 128), (777, 64, 256, 128), (100, 160, 640, 128), (31, 32, 32, 32),
                                        (640, 320, 128, 64), (130, 96, 160, 64), (200, 160, 640, 96), (26464, 160, 640, 2048)])
def test_gemm_tn(ctx, M, NY, NX, mc):
    lib, L, dev = ctx
    g = torch.Generator().manual_seed(M + NY)
    Y = torch.randn(M, NY, generator=g).to(dev).bfloat16()
    X = torch.randn(M, NX, generator=g).to(dev).bfloat16()
    ref = Y.float().t() @ X.float()
    dW = torch.zeros(NY, NX, device=dev)
    L.check(lib.v1t_gemm_tn(Y.data_ptr(), NY, X.data_ptr(), NX, M, NY, NX, dW.data_ptr(), NX, mc, L.stream()))
    check_rel("test_gemm_tn:2", dW.cpu(), ref.cpu(), 5e-6)  # fp32 atomics: order-dependent last bits only
    L.check(lib.v1t_gemm_tn(Y.data_ptr(), NY, X.data_ptr(), NX, M, NY, NX, dW.data_ptr(), NX, mc, L.stream()))
    check_rel("test_gemm_tn:3", dW.cpu(), 2 * ref.cpu(), 5e-6)  # accumulates (+=)
    # slab + reduce form (what the ViT backward uses); same result, deterministic
    nb = lib.v1t_gemm_tn_slab_bytes(M, NY, NX, mc)
    slab = torch.empty(max(nb, 4) // 4, device=dev)
    outs = []
    for _ in range(2):
        dW2 = torch.zeros(NY, NX, device=dev)
        L.check(lib.v1t_gemm_tn_slab(Y.data_ptr(), NY, X.data_ptr(), NX, M, NY, NX, dW2.data_ptr(), NX, mc, slab.data_ptr(), nb, L.stream()))
        outs.append(dW2.cpu())
    check_rel("test_gemm_tn:4", outs[0], ref.cpu(), 5e-6)
    if nb > 0:
        assert torch.equal(outs[0], outs[1])


def _attn_ref(qkv, B, H, T, DP, scale, mask=None, p=0.0, diag=False):
    q, k, v = qkv.float().view(B, T, 3, H, DP).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * scale.view(1, -1, 1, 1)
    if diag:
        s = s.masked_fill(torch.eye(T, dtype=torch.bool, device=s.device), -torch.finfo(torch.float32).max)
    a = torch.softmax(s, -1)
    if mask is not None:
        a = a * mask.view(B, H, T, T).float() / (1 - p)
    return (a @ v).permute(0, 2, 1, 3).reshape(B * T, H * DP)


@pytest.mark.parametrize("B,H,T,DP,p,lsa", [(2, 4, 1654, 160, 0.0, False), (2, 4, 1654, 64, 0.0, False), (1, 2, 100, 160, 0.0, False), (1, 1, 1, 32, 0.0, False),
                                              (2, 3, 333, 64, 0.25, False), (1, 4, 1654, 160, 0.2544, False), (2, 2, 257, 96, 0.1, True), (1, 2, 130, 128, 0.0, True),
                                              # T % 128 <= 96 with head dim >= 128: the dQ GEMM's last 32-key tiles lie wholly beyond T (round 4: a negative row clamp there
                                              # became a 4 GB lane offset - found by tests/test_gpu_longseq.py at T = 34 114; T = 1654 has no such tile)
                                              (1, 2, 200, 160, 0.1, False), (2, 1, 130, 128, 0.0, False), (1, 2, 436, 160, 0.0, False)])
def test_attention_forward_backward(ctx, B, H, T, DP, p, lsa):
    lib, L, dev = ctx
    g = torch.Generator().manual_seed(B * 1000 + T + DP)
    qkv = (torch.randn(B * T, 3 * H * DP, generator=g) * 0.7).to(dev).bfloat16().requires_grad_(True)
    scale = (torch.full((H,), DP ** -0.5) * (1 + 0.2 * torch.randn(H, generator=g))).to(dev) if lsa else torch.tensor([DP ** -0.5], device=dev)
    scale_r = scale.clone().requires_grad_(True)
    o = torch.empty(B * T, H * DP, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B, H, T, device=dev)
    seed, sid = 4242, 16  # attention-P streams are 8*block + 0 (include/v1t_amd.h)
    L.check(lib.v1t_attention_forward(qkv.data_ptr(), B, H, T, DP, scale.data_ptr(), int(lsa), int(lsa), p, seed, sid, o.data_ptr(), lse.data_ptr(), L.stream()))
    # bit-reproducible (guards the hand-placed MFMA -> VALU wait states: a missing one shows up as run-to-run noise)
    o_again, lse_again = torch.empty_like(o), torch.empty_like(lse)
    L.check(lib.v1t_attention_forward(qkv.data_ptr(), B, H, T, DP, scale.data_ptr(), int(lsa), int(lsa), p, seed, sid, o_again.data_ptr(), lse_again.data_ptr(), L.stream()))
    assert torch.equal(o, o_again) and torch.equal(lse, lse_again)
    mask = None
    if p > 0:
        mask = torch.empty(B * H * T, T, device=dev, dtype=torch.uint8)
        L.check(lib.v1t_dropout_mask(seed, sid, p, B * H * T, T, mask.data_ptr(), L.stream()))
        if mask.numel() > 10000:
            assert abs(float(mask.float().mean()) - (1 - float(lib.v1t_attention_dropout_rate(p)))) < 5e-3  # keep rate of the counter-based mask
    p_eff = float(lib.v1t_attention_dropout_rate(p))  # round(65536 p) / 65536 (include/v1t_amd.h)
    ref = _attn_ref(qkv, B, H, T, DP, scale_r if lsa else scale_r.expand(H), mask, p_eff, diag=lsa)
    assert rel_to_max(o.float().cpu(), ref.detach().cpu()) < 1e-2  # bf16 P and bf16 output
    dO = (torch.randn(B * T, H * DP, generator=g) * 0.5).to(dev).bfloat16()
    gq, gs = torch.autograd.grad(ref, (qkv, scale_r), dO.float())
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B, H, T, device=dev)
    dscale = torch.zeros(H if lsa else 1, device=dev)
    L.check(lib.v1t_attention_backward(qkv.data_ptr(), o.data_ptr(), dO.data_ptr(), lse.data_ptr(), B, H, T, DP, scale.data_ptr(), int(lsa), int(lsa), p, seed,
                                       sid, delta.data_ptr(), dqkv.data_ptr(), dscale.data_ptr(), L.stream()))
    # bit-reproducible backward as well (no float atomics on dq / dk / dv; hand-placed instruction order and hazards)
    dqkv_again = torch.zeros_like(dqkv)
    L.check(lib.v1t_attention_backward(qkv.data_ptr(), o.data_ptr(), dO.data_ptr(), lse.data_ptr(), B, H, T, DP, scale.data_ptr(), int(lsa), int(lsa), p, seed,
                                       sid, delta.data_ptr(), dqkv_again.data_ptr(), torch.zeros_like(dscale).data_ptr(), L.stream()))
    assert torch.equal(dqkv, dqkv_again)
    gq, d = gq.view(B * T, 3, H * DP), dqkv.float().view(B * T, 3, H * DP)
    for i, nm in enumerate("qkv"):
        if float(gq[:, i].abs().max()) > 0:
            check_rel(f"test_attention_forward_backward:" + str(nm), d[:, i].cpu(), gq[:, i].cpu(), 1.2e-2)  # bf16 P / dS / outputs (worst measured 7.8e-3, GPUTEST r05)
    if lsa and T > 1:  # the scale is a learnable parameter only with LSA (vit.py:235-239)
        check_rel("test_attention_forward_backward:6", dscale.cpu(), gs.cpu(), 1.2e-2)
    if not lsa:
        # materialised-dS' path (producer / consumer dK/dV kernel writes dS', dQ = dS' . K as a GEMM): same bf16 products as
        # the recompute path, summed in another order
        nb = int(lib.v1t_attention_backward_ws_bytes(B, H, T))
        ws = torch.full((nb,), 0xFF, dtype=torch.uint8, device=dev)  # NaN patterns: every element read must have been written
        d2 = torch.zeros_like(dqkv)
        for out in (d2, torch.zeros_like(dqkv)):
            L.check(lib.v1t_attention_backward_ws(qkv.data_ptr(), o.data_ptr(), dO.data_ptr(), lse.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, seed, sid,
                                                  delta.data_ptr(), out.data_ptr(), None, ws.data_ptr(), nb, L.stream()))
        assert torch.equal(d2, out)  # bit-reproducible
        e = d2.float().view(B * T, 3, H * DP)
        for i, nm in enumerate("qkv"):
            if float(gq[:, i].abs().max()) > 0:
                check_rel(f"test_attention_forward_backward:" + str(nm), e[:, i].cpu(), gq[:, i].cpu(), 1.2e-2)
                check_rel(f"test_attention_forward_backward:" + str(nm), e[:, i].cpu(), d[:, i].cpu(), 1e-2)


@pytest.mark.parametrize("B,C,H,W,N", [(3, 155, 29, 57, 1000), (2, 64, 29, 57, 257), (1, 40, 15, 29, 3), (2, 155, 5, 7, 900),
                                       # edge shapes (round 4): one cell / one neuron / one channel, channel counts around the 64-lane passes, 33 images, a map of
                                       # exactly 4096 cells (the sorted backward's LDS histogram) and one beyond it (atomic form)
                                       (1, 1, 1, 1, 1), (33, 63, 2, 3, 7), (2, 65, 3, 2, 130), (1, 160, 64, 64, 300), (2, 129, 65, 64, 50), (17, 256, 4, 4, 64)])
def test_readout(ctx, B, C, H, W, N):
    from oracle import v1t_oracle as O

    lib, L, dev = ctx
    g = torch.Generator().manual_seed(N)
    z = torch.randn(B, C, H, W, generator=g).to(dev)
    grid = (torch.rand(B, N, 2, generator=g) * 2.4 - 1.2).to(dev)  # includes points outside [-1, 1] (zero padding)
    grid[0, 0] = torch.tensor([-1.0, -1.0])
    grid[0, min(1, N - 1)] = torch.tensor([1.0, 1.0])
    FS = (C + 31) // 32 * 32
    feat = torch.zeros(N, FS, device=dev)
    feat[:, :C] = torch.randn(N, C, generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    zr, gr, fr = z.clone().requires_grad_(True), grid.clone().requires_grad_(True), feat[:, :C].clone().requires_grad_(True)
    ref = (O.bilinear_sample(zr, gr) * fr.t()[None]).sum(1) + bias
    zl = z.permute(0, 2, 3, 1).contiguous()
    out = torch.empty(B, N, device=dev)
    L.check(lib.v1t_gaussian2d_forward(zl.data_ptr(), H * W * C, C, B, C, H, W, N, grid.data_ptr(), feat.data_ptr(), FS, bias.data_ptr(), out.data_ptr(), L.stream()))
    check_rel("test_readout:9", out.cpu(), ref.detach().cpu(), 2e-6)
    go = torch.randn(B, N, generator=g).to(dev)
    gz, gg, gf = torch.autograd.grad(ref, (zr, gr, fr), go)
    dz, dgrid, dfeat, dbias = torch.zeros_like(zl), torch.empty_like(grid), torch.zeros_like(feat), torch.zeros_like(bias)
    L.check(lib.v1t_gaussian2d_backward(zl.data_ptr(), H * W * C, C, B, C, H, W, N, grid.data_ptr(), feat.data_ptr(), FS, go.data_ptr(), dz.data_ptr(), H * W * C, C,
                                        dgrid.data_ptr(), dfeat.data_ptr(), dbias.data_ptr(), L.stream()))
    check_rel("test_readout:10", dz.permute(0, 3, 1, 2).cpu(), gz.cpu(), 5e-6)
    check_rel("test_readout:11", dgrid.cpu(), gg.cpu(), 5e-6)
    check_rel("test_readout:12", dfeat[:, :C].cpu(), gf.cpu(), 5e-6)
    check_rel("test_readout:13", dbias.cpu(), go.sum(0).cpu(), 5e-6)
    # scratch form: dz gathered through the inverted tap index (the 5x7 map overflows the 64-entry cell lists,
    # which exercises the atomic overflow path as well)
    nb = lib.v1t_gaussian2d_backward_ws_bytes(B, H, W, N)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    dz2, dgrid2, dfeat2, dbias2 = torch.zeros_like(zl), torch.empty_like(grid), torch.zeros_like(feat), torch.zeros_like(bias)
    L.check(lib.v1t_gaussian2d_backward_ws(zl.data_ptr(), H * W * C, C, B, C, H, W, N, grid.data_ptr(), feat.data_ptr(), FS, go.data_ptr(), dz2.data_ptr(), H * W * C, C,
                                           dgrid2.data_ptr(), dfeat2.data_ptr(), dbias2.data_ptr(), ws.data_ptr(), nb, L.stream()))
    check_rel("test_readout:14", dz2.permute(0, 3, 1, 2).cpu(), gz.cpu(), 5e-6)
    check_rel("test_readout:15", dgrid2.cpu(), gg.cpu(), 5e-6)
    check_rel("test_readout:16", dfeat2[:, :C].cpu(), gf.cpu(), 5e-6)
    check_rel("test_readout:17", dbias2.cpu(), go.sum(0).cpu(), 5e-6)


@pytest.mark.parametrize("B,T,D,DP", [(2, 1654, 64, 64), (2, 1654, 155, 160), (3, 100, 40, 64)])
def test_layernorm(ctx, B, T, D, DP):
    from oracle import v1t_oracle as O

    lib, L, dev = ctx
    g = torch.Generator().manual_seed(D)
    x = torch.zeros(B, T, DP)
    x[:, :, :D] = torch.randn(B, T, D, generator=g)
    inj = torch.zeros(B, DP)
    inj[:, :D] = torch.randn(B, D, generator=g)
    gd, bd = (1 + 0.1 * torch.randn(D, generator=g)).to(dev), (0.1 * torch.randn(D, generator=g)).to(dev)
    xd, injd = x.to(dev), inj.to(dev)
    xout, z = torch.empty_like(xd), torch.empty(B, T, DP, device=dev, dtype=torch.bfloat16)
    mean, rstd = torch.empty(B * T, device=dev), torch.empty(B * T, device=dev)
    L.check(lib.v1t_layernorm_forward(xd.data_ptr(), injd.data_ptr(), xout.data_ptr(), gd.data_ptr(), bd.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), B, T, D, DP, 1e-5, L.stream()))
    xr = (xd[:, :, :D] + injd[:, None, :D]).requires_grad_(True)
    gam = gd.clone().requires_grad_(True)
    ref = O.layer_norm(xr, gam, bd)
    check_rel("test_layernorm:18", z[:, :, :D].float().cpu(), ref.detach().cpu(), 4e-3)  # bf16 output
    assert float(z[:, :, D:].float().abs().max()) == 0.0 if DP > D else True
    dz, gin = torch.zeros(B, T, DP), torch.zeros(B, T, DP)
    dz[:, :, :D], gin[:, :, :D] = torch.randn(B, T, D, generator=g), torch.randn(B, T, D, generator=g)
    dzd, gind = dz.to(dev), gin.to(dev)
    gout = torch.empty_like(gind)
    dgamma, dbeta, dinj, dbn = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(B, DP, device=dev), torch.zeros(D, device=dev)
    dyn = torch.empty(B, T, DP, device=dev, dtype=torch.bfloat16)
    L.check(lib.v1t_layernorm_backward(dzd.data_ptr(), xout.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gd.data_ptr(), gind.data_ptr(), gout.data_ptr(), dgamma.data_ptr(),
                                       dbeta.data_ptr(), dinj.data_ptr(), dyn.data_ptr(), dbn.data_ptr(), B, T, D, DP, L.stream()))
    gx, gg = torch.autograd.grad(ref, (xr, gam), dzd[:, :, :D])
    full = gx + gind[:, :, :D]
    check_rel("test_layernorm:19", gout[:, :, :D].cpu(), full.cpu(), 2e-6)
    check_rel("test_layernorm:20", dgamma.cpu(), gg.cpu(), 5e-6)
    check_rel("test_layernorm:21", dbeta.cpu(), dzd[:, :, :D].sum((0, 1)).cpu(), 5e-6)
    check_rel("test_layernorm:22", dinj[:, :D].cpu(), full.sum(1).cpu(), 5e-6)
    check_rel("test_layernorm:23", dyn[:, :, :D].float().cpu(), full.cpu(), 4e-3)


def test_adamw_l1_and_loss(ctx):
    from oracle import v1t_oracle as O

    lib, L, dev = ctx
    g = torch.Generator().manual_seed(3)
    n = 100003
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    p, grad, m, v = p0.to(dev), gr.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    params, st = {"p": p0.clone()}, {}
    for step in (1, 2, 3):
        L.check(lib.v1t_adamw_step(p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1.647e-3, 0.9, 0.9999, 1e-8, 0.0, step, 0.5, 0, L.stream()))
        O.adamw_step(params, {"p": gr + 0.5 * torch.sign(params["p"])}, st, step=step, lr=1.647e-3)
    check_rel("test_adamw_l1_and_loss:24", p.cpu(), params["p"], 1e-6)
    out = torch.zeros((), device=dev)
    L.check(lib.v1t_l1_sum(p.data_ptr(), n, 0.25, out.data_ptr(), L.stream()))
    assert abs(float(out) - 0.25 * float(p.abs().sum())) / float(out) < 1e-5
    # ELU1 + Poisson
    u = (torch.randn(16, 1000, generator=g) * 3).to(dev)
    u[0, :4] = torch.tensor([-20.0, -17.0, 0.0, 15.0], device=dev)
    y = torch.empty(16, 1000).exponential_(1.0, generator=g).to(dev)
    ur = u.clone().requires_grad_(True)
    yh_ref = O.elu1(ur)
    loss_ref = O.poisson_loss(y, yh_ref, 4500.0, 16)
    (gu,) = torch.autograd.grad(loss_ref, ur)
    yh, du, loss = torch.empty_like(u), torch.empty_like(u), torch.zeros((), device=dev)
    L.check(lib.v1t_elu1_poisson(u.data_ptr(), y.data_ptr(), u.numel(), math.sqrt(4500.0 / 16), 1.0, yh.data_ptr(), du.data_ptr(), loss.data_ptr(), L.stream()))
    assert torch.allclose(yh, yh_ref.detach(), rtol=1e-6, atol=1.2e-7)  # incl. the expm1 quantisation near 0
    assert abs(float(loss) - float(loss_ref)) / abs(float(loss_ref)) < 1e-5
    check_rel("test_adamw_l1_and_loss:25", du.cpu(), gu.cpu(), 1e-5)


def test_elu1_poisson_edge_vs_reference_golden(ctx, golden):
    lib, L, dev = ctx
    u = torch.from_numpy(golden["elu_edge/u"]).to(dev)
    y = torch.from_numpy(golden["elu_edge/y_true"]).to(dev)
    yh, du, loss = torch.empty_like(u), torch.empty_like(u), torch.zeros((), device=dev)
    L.check(lib.v1t_elu1_poisson(u.data_ptr(), y.data_ptr(), u.numel(), math.sqrt(4500.0 / 16), 1.0, yh.data_ptr(), du.data_ptr(), loss.data_ptr(), L.stream()))
    assert torch.allclose(yh.cpu(), torch.from_numpy(golden["elu_edge/yhat"]), rtol=1e-6, atol=1.2e-7)
    assert abs(float(loss) - float(golden["elu_edge/loss"])) <= 1e-5 * abs(float(golden["elu_edge/loss"]))
    check_rel("test_elu1_poisson_edge_vs_reference_golden:26", du.cpu(), golden["elu_edge/du"], 1e-5)


@pytest.mark.parametrize("shape,out", [((2, 1, 144, 256), (36, 64)), ((3, 2, 37, 50), (36, 64)), ((1, 1, 36, 64), (36, 64)), ((2, 1, 20, 30), (45, 77))])
def test_resize_bilinear(ctx, shape, out):
    """ImageCropper resize (image_cropper.py:96-99) against the oracle's explicit half-pixel taps (pinned to
    F.interpolate / the reference by tests/golden: resize/out_sample)."""
    from oracle import v1t_oracle as O

    lib, L, dev = ctx
    x = torch.from_numpy(np.random.default_rng(5).standard_normal(shape).astype(np.float32))
    ref = O.resize_bilinear(x, out)
    xd = x.to(dev)
    y = torch.empty(shape[0], shape[1], *out, device=dev)
    L.check(lib.v1t_resize_bilinear(xd.data_ptr(), shape[0] * shape[1], shape[2], shape[3], y.data_ptr(), out[0], out[1], L.stream()))
    assert float((y.cpu() - ref).abs().max()) < 1e-5  # fp32 tap weights: (dst + 0.5) * scale - 0.5 rounds differently at non-integer scales


@pytest.mark.parametrize("B,N,gd,sample", [(16, 8000, 2, True), (5, 333, 3, True), (40, 1000, 2, True), (3, 64, 0, True), (4, 130, 2, False)])
def test_readout_grid_backward_paths(ctx, B, N, gd, sample):
    """Grid backward (gaussian2d.py:188-235, 265-268): the two-stage workspace path and the atomics path against torch
    fp32 autograd of mu = tanh(W2 . ELU(W0 . src + b0) + b2), grid = clamp(sigma . eps + mu, -1, 1) + shift."""
    lib, L, dev = ctx
    g = torch.Generator().manual_seed(B * 1000 + N)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    src, W0, b0, W2, b2 = r(N, max(gd, 1)), r(30, max(gd, 1), sc=0.5), r(30, sc=0.3), r(2, 30, sc=0.3), r(2, sc=0.1)
    mu_free, sigma, eps, shift, dgrid = r(N, 2, sc=0.6), r(N, 2, 2, sc=0.4), r(B, N, 2), r(B, 2, sc=0.1), r(B, N, 2)
    leaves = [t_.clone().requires_grad_(True) for t_ in (W0, b0, W2, b2, mu_free, sigma, shift)]
    lW0, lb0, lW2, lb2, lmu, lsig, lsh = leaves
    mu = torch.tanh(torch.nn.functional.elu(src[:, :gd] @ lW0[:, :gd].T + lb0) @ lW2.T + lb2) if gd > 0 else lmu
    grid = (torch.einsum("ncd,bnd->bnc", lsig, eps) + mu[None]) if sample else mu[None].expand(B, -1, -1)
    grid = grid.clamp(-1, 1) + lsh[:, None, :]
    grid.backward(dgrid)
    d = lambda t_: t_.contiguous().to(dev)
    srcd, W0d, b0d, W2d, b2d = d(src[:, :max(gd, 1)]), d(W0[:, :max(gd, 1)]), d(b0), d(W2), d(b2)
    mud, sigd, epsd, dgd = d(mu_free), d(sigma), d(eps), d(dgrid)
    for use_ws in (True, False):
        o = {k: torch.zeros(*s, device=dev) for k, s in (("dW0", (30, max(gd, 1))), ("db0", (30,)), ("dW2", (2, 30)), ("db2", (2,)), ("dmu", (N, 2)),
                                                           ("dsig", (N, 2, 2)), ("dsh", (B, 2)))}
        nbytes = int(lib.v1t_readout_grid_backward_ws_bytes(B, N)) if use_ws else 0
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        for rep in range(2):  # outputs accumulate (+=): the second call doubles the predictor / shift gradients
            L.check(lib.v1t_readout_grid_backward_ws(B, N, gd, srcd.data_ptr() if gd else None, *[(x.data_ptr() if gd else None) for x in (W0d, b0d, W2d, b2d)],
                                                     None if gd else mud.data_ptr(), sigd.data_ptr(), epsd.data_ptr() if sample else None, dgd.data_ptr(),
                                                     *[(o[k].data_ptr() if gd else None) for k in ("dW0", "db0", "dW2", "db2")],
                                                     None if gd else o["dmu"].data_ptr(), o["dsig"].data_ptr() if sample else None, o["dsh"].data_ptr(),
                                                     ws.data_ptr() if use_ws else None, nbytes, L.stream()))
        torch.cuda.synchronize()
        tag = f"ws={use_ws}"
        check_rel(f"test_readout_grid_backward_paths:" + str(tag), o["dsh"].cpu(), 2 * lsh.grad, 2e-5)
        if sample:
            check_rel(f"test_readout_grid_backward_paths:" + str(tag), o["dsig"].cpu(), lsig.grad, 2e-5)
        if gd:
            check_rel(f"test_readout_grid_backward_paths:" + str(tag), o["dW0"].cpu(), 2 * lW0.grad[:, :gd], 5e-5)
            check_rel(f"test_readout_grid_backward_paths:" + str(tag), o["db0"].cpu(), 2 * lb0.grad, 5e-5)
            check_rel(f"test_readout_grid_backward_paths:" + str(tag), o["dW2"].cpu(), 2 * lW2.grad, 5e-5)
            check_rel(f"test_readout_grid_backward_paths:" + str(tag), o["db2"].cpu(), 2 * lb2.grad, 5e-5)
        else:
            check_rel(f"test_readout_grid_backward_paths:" + str(tag), o["dmu"].cpu(), lmu.grad, 2e-5)


def test_attention_forward_role_variant_subprocess():
    """The 16-wave S-wave / PV-wave forward (attn_fwd3_kernel, opt-in through V1T_ATTN_FWD_V3: the switch is read once per process)
    against the default forward kernel on the same inputs, dropout on: same masks, same math, other summation order."""
    import os
    import subprocess
    import sys

    code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from v1t_amd import lib as L
lib = L.load()
dev = torch.device("cuda:0")
B, H, T, DP = 3, 4, 1654, 160
g = torch.Generator().manual_seed(3)
qkv = (torch.randn(B * T, 3 * H * DP, generator=g) * 0.7).to(dev).bfloat16()
qkv[5, :DP] *= 30  # one query with very large scores: forces the reference-rescale path (T13) in some tile
scale = torch.tensor([DP ** -0.5], device=dev)
o = torch.empty(B * T, H * DP, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, H, T, device=dev)
L.check(lib.v1t_attention_forward(qkv.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, 0.2544, 77, 8, o.data_ptr(), lse.data_ptr(), L.stream()))
torch.cuda.synchronize()
torch.save({"o": o.float().cpu(), "lse": lse.cpu()}, sys.argv[1])
'''
    outs = []
    for v2 in (False, True):
        path = f"/tmp/v1t_fwd_{int(v2)}.pt"
        env = exp_env(V1T_ATTN_FWD_V3=1) if v2 else product_env()
        r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(path))
    a, b = outs
    assert bool(torch.isfinite(b["o"]).all())
    check_rel("test_attention_forward_role_variant_subprocess:34", b["o"], a["o"], 1e-2)
    assert float((b["lse"] - a["lse"]).abs().max()) < 2e-2


@pytest.mark.parametrize("emb,images", [(64, 3), (155, 2), (96, 5)])
def test_fused_layernorm_gemm_equals_two_kernels(emb, images):
    """LN1 -> QKV and LN2 -> FC1 as one A-stationary launch (gemm.hip ln_gemm_kernel, 256-row workgroups) against ln_fwd + gemm_nt
    (V1T_LN_FUSE=0, read once per process: two subprocesses): core output, loss gradient of every core parameter. Row counts that
    are no multiple of 128 / 256 (the second 128-row half of the last workgroup is empty or ragged)."""
    import os
    import subprocess
    import sys

    code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from oracle import v1t_oracle as O
from oracle import weights as W
from tests.helpers import build_native_model
emb, images = int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
cfg = O.Config(num_blocks=2, emb_dim=emb, mlp_dim=2 * emb + 24, num_heads=4, mouse_ids=("A",), num_neurons={"A": 40})
sd = W.make_state_dict(cfg, 5)
model, _ = build_native_model(cfg, sd, dev)
model.train(False)
b = {k: v.to(dev) for k, v in W.make_batch(cfg, "A", images, 5).items()}
model.core.prepare()
model.core._arena.attach_grads()
model.core._arena.grad.zero_()
u = model(inputs=b["image"], mouse_id="A", behaviors=b["behavior"], pupil_centers=b["pupil_center"], activate=False)[0]
(u * torch.linspace(-1, 1, u.numel(), device=dev).view_as(u)).sum().backward()
torch.cuda.synchronize()
torch.save({"u": u.detach().cpu(), "g": model.core._arena.grad.cpu()}, sys.argv[1])
'''
    outs = []
    for fuse in ("0", "1"):
        path = f"/tmp/v1t_lnfuse_{fuse}.pt"
        env = exp_env(V1T_LN_FUSE=0) if fuse == "0" else product_env()
        r = subprocess.run([sys.executable, "-c", code, path, str(emb), str(images)], env=env, capture_output=True, text=True,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(path))
    a, b = outs
    assert bool(torch.isfinite(b["u"]).all()) and bool(torch.isfinite(b["g"]).all())
    # same arithmetic up to the summation order of the row statistics (a rounding flip of a 16-bit operand now and then)
    check_rel("test_fused_layernorm_gemm_equals_two_kernels:35", b["u"], a["u"], 1e-3)
    check_rel("test_fused_layernorm_gemm_equals_two_kernels:36", b["g"], a["g"], 5e-3)


@pytest.mark.parametrize("switch,tol_g", [("V1T_LNBWD_UNFUSED", 2e-4), ("V1T_KEEP_BF16_PLANES", 2e-3), ("V1T_DELTA_UNFUSED", 2e-4)])
@pytest.mark.parametrize("emb,images,beh", [(155, 2, 3), (64, 3, 0), (96, 5, 3)])
def test_backward_fusions_equal_their_unfused_forms(switch, tol_g, emb, images, beh):
    """Round-3 backward fusions against the kernels they replace, same process image apart from one dev switch (read once per
    process: two subprocesses), training mode (dropout masks are counter-based: identical in both runs), ragged row counts:
      V1T_LNBWD_UNFUSED    - gemm_lnbwd_kernel (dX GEMM + LayerNorm backward on the accumulators) vs gemm_nt<EPI_F32> + ln_bwd_kernel:
                             same arithmetic, other summation order of the row / column sums (fp32);
      V1T_KEEP_BF16_PLANES - weight-gradient GEMM and row constants reading the forward's fp16 planes of the attention output / GELU
                             output (converted to bf16 fragment by fragment) vs the bf16 planes: the X operand is bf16(fp16(x))
                             instead of bf16(x), a second rounding of at most half a bf16 ulp on some elements.
      V1T_DELTA_UNFUSED    - the attention backward's row constants (delta = rowsum(dO * O) per head, -lse) out of the dO GEMM's epilogue
                             (gemm.h RowDotArgs) vs attn_delta2_kernel: same bf16-rounded dO, other summation order.
    Compared: every core gradient (one flat arena) and the loss value."""
    import os
    import subprocess
    import sys

    code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from oracle import v1t_oracle as O
from oracle import weights as W
from tests.helpers import build_native_model
emb, images, beh = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda:0")
cfg = O.Config(num_blocks=2, emb_dim=emb, mlp_dim=(488 if emb == 155 else 2 * emb + 24), num_heads=4, behavior_mode=beh, mouse_ids=("A",), num_neurons={"A": 40},
               p_dropout=0.1, t_dropout=0.2)
sd = W.make_state_dict(cfg, 5)
model, _ = build_native_model(cfg, sd, dev)
model.train(True)
b = {k: v.to(dev) for k, v in W.make_batch(cfg, "A", images, 5).items()}
model.core.prepare()
model.core._arena.attach_grads()
model.core._arena.grad.zero_()
torch.manual_seed(11)
u = model(inputs=b["image"], mouse_id="A", behaviors=b["behavior"], pupil_centers=b["pupil_center"], activate=False)[0]
loss = (u * torch.linspace(-1, 1, u.numel(), device=dev).view_as(u)).sum()
loss.backward()
torch.cuda.synchronize()
torch.save({"u": u.detach().cpu(), "g": model.core._arena.grad.cpu()}, sys.argv[1])
'''
    outs = []
    for on in (False, True):
        path = f"/tmp/v1t_{switch}_{int(on)}.pt"
        env = exp_env(**{switch: 1}) if on else product_env()
        r = subprocess.run([sys.executable, "-c", code, path, str(emb), str(images), str(beh)], env=env, capture_output=True, text=True,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(path))
    a, b = outs
    assert bool(torch.isfinite(a["u"]).all()) and bool(torch.isfinite(a["g"]).all()) and float(a["g"].abs().max()) > 0
    assert torch.equal(a["u"], b["u"])  # neither switch touches the forward's values
    check_rel(f"test_backward_fusions_equal_their_unfused_forms:{switch}:{emb}", a["g"], b["g"], tol_g)


@pytest.mark.parametrize("B,T", [(2, 70), (1, 300), (3, 129)])
def test_rollout_matmul_vs_fp64(B, T):
    """v1t_rollout_matmul (one step of the full rollout chain, X <- X . A_hat^T, split-bf16 MFMA products) against fp64 torch:
    ragged T (no multiple of the 128 x 128 tile or the 32-wide K tile), identity start (Xin = NULL), two chained steps."""
    import ctypes as C

    from v1t_amd import lib as L

    lib = L.load()
    dev = torch.device("cuda:0")
    TP = (T + 3) // 4 * 4
    g = torch.Generator().manual_seed(T)
    steps = []
    for _ in range(2):
        A = torch.rand(B, T, TP, generator=g) * (2.0 / T)
        A[:, :, T:] = 7.0  # pad columns must be ignored
        steps.append(A.to(dev))
    ref = None
    cur = None
    bufs = [torch.full((B, T, TP), float("nan"), device=dev) for _ in range(2)]
    for k, A in enumerate(steps):
        a64 = A[:, :, :T].double().cpu() + torch.eye(T, dtype=torch.float64)
        rowsum = a64.sum(-1)
        ahat = a64 / rowsum[..., None]
        ref = ahat if ref is None else ahat @ ref
        rs = rowsum.float().to(dev)
        out = bufs[k & 1]
        L.check(lib.v1t_rollout_matmul(A.data_ptr(), rs.data_ptr(), L.ptr(cur), out.data_ptr(), B, T, TP, L.stream()), "rollout_matmul")
        cur = out
    got = cur[:, :, :T].transpose(1, 2).double().cpu()  # X = result^T
    assert bool(torch.isfinite(cur).all())
    assert float((cur[:, :, T:]).abs().max()) == 0.0 if TP > T else True
    check_rel("test_rollout_matmul_vs_fp64:37", got, ref, 5e-5)  # split-bf16: ~2^-16 of the largest entry
    assert L.load().v1t_rollout_matmul(steps[0].data_ptr(), rs.data_ptr(), cur.data_ptr(), cur.data_ptr(), B, T, TP, L.stream()) != 0  # aliasing refused


def test_attention_dropout_rate_quantisation():
    """The attention-P dropout runs at round(65536 p) / 65536 (byte decisions against a per-tile dithered threshold): the default 0.2544
    within 2e-5 relative, a sweep near 0 follows p down to 2^-17, below which the site is OFF; the top is 65535 / 65536."""
    from v1t_amd import lib as L

    lib = L.load()
    assert float(lib.v1t_attention_dropout_rate(0.0)) == 0.0
    assert float(lib.v1t_attention_dropout_rate(2.0 ** -18)) == 0.0
    for p in (1e-4, 0.001, 0.003, 0.01, 0.1, 0.2544, 0.5, 0.9):
        got = float(lib.v1t_attention_dropout_rate(p))
        assert abs(got - p) <= 2.0 ** -17 + 1e-9, (p, got)
    assert abs(float(lib.v1t_attention_dropout_rate(0.2544)) - 16672.0 / 65536.0) < 1e-7
    assert float(lib.v1t_attention_dropout_rate(0.99999999)) == 65535.0 / 65536.0


@pytest.mark.parametrize("p", [0.003, 0.2544])
def test_attention_dropout_mask_statistics(ctx, p):
    """The exported mask of the attention-P site (v1t_dropout_mask on an attention stream = the decisions the kernels take, compared bit
    for bit by the parity tests) keeps 1 - p of the elements - also at a rate the fixed byte threshold of rounds 1-4 could not express - with
    no row / column structure and no correlation between neighbours."""
    lib, L, dev = ctx
    BH, T = 8, 1654
    m = torch.empty(BH * T, T, dtype=torch.uint8, device=dev)
    L.check(lib.v1t_dropout_mask(1234, 8, p, BH * T, T, m.data_ptr(), L.stream()))
    k = m.view(BH, T, T).float()
    p_eff = float(lib.v1t_attention_dropout_rate(p))
    n = k.numel()
    sd = (p_eff * (1 - p_eff) / n) ** 0.5
    assert abs(float(1 - k.mean()) - p_eff) < 5 * sd + 1e-6
    sd_line = (p_eff * (1 - p_eff) / T) ** 0.5  # a row's / column's drop rate: binomial over T elements
    assert 0.85 * sd_line < float(k.mean(2).std()) < 1.15 * sd_line and 0.85 * sd_line < float(k.mean(1).std()) < 1.15 * sd_line
    c = k - k.mean()
    var = float((c * c).mean())
    for a_, b_ in ((c[:, :, :-1], c[:, :, 1:]), (c[:, :-1], c[:, 1:]), (c[:, :-1, :-1], c[:, 1:, 1:]), (c[:-1], c[1:])):
        assert abs(float((a_ * b_).mean()) / var) < 0.01


def test_attention_forward_extreme_scores(ctx):
    """Online softmax of the forward kernel with its deferred rescale (the running maximum moves only when a score exceeds it by
    2^6): queries whose scores are all huge of either sign, whose FIRST key tile is huge against a moderate rest, and a huge score
    late in the row - against fp32 torch, forward output and log-sum-exp."""
    lib, L, dev = ctx
    B, H, T, DP = 1, 2, 200, 160
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B * T, 3, H, DP, generator=g) * 0.7
    x[3, 0] *= 40.0                     # query 3: huge scores of either sign
    x[7, 0, 0] = -25.0 * x[:, 1, 0].mean(0).sign() * 1.0   # query 7 (head 0): strongly anti-aligned with the mean key
    x[:32, 1, 1] *= 25.0                # head 1: the first 32 keys are huge -> first-tile scores dominate or vanish
    x[150, 1, 0] *= 50.0                # head 0: one huge key late in the row
    qkv = x.reshape(B * T, 3 * H * DP).to(dev).bfloat16()
    scale = torch.tensor([DP ** -0.5], device=dev)
    o = torch.empty(B * T, H * DP, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B, H, T, device=dev)
    L.check(lib.v1t_attention_forward(qkv.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, 0.0, 1, 8, o.data_ptr(), lse.data_ptr(), L.stream()))
    assert bool(torch.isfinite(o.float()).all()) and bool(torch.isfinite(lse).all())
    q, k, v = (qkv.float().view(B, T, 3, H, DP)[:, :, i].permute(0, 2, 1, 3) for i in range(3))  # (B, H, T, DP)
    s = (q @ k.transpose(-1, -2)) * float(scale)
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * T, H * DP)
    ref_lse2 = torch.logsumexp(s, -1) * 1.4426950408889634
    check_rel("test_attention_forward_extreme_scores:38", o.float().cpu(), ref.cpu(), 1.5e-2)
    # scores of ~1e3 in log2 units
    assert float(((lse - ref_lse2).abs() / (1.0 + 4e-3 * ref_lse2.abs())).max()) < 1.0


@pytest.mark.parametrize("p", [0.0, 0.2544])
def test_attention_backward_extreme_scores(ctx, p):
    """The default backward pair - `attn_bwd_dkv2` (S' = c q.k - lse folded into the accumulator, dS' materialised as bf16) and `attn_bwd_dq2`
    - at the production shape (head dim 160, T = 1654) in the regime of trained weights: score std ~ 6 (peaked rows: the largest
    probability of a row ~ 0.3-1), single queries / keys scaled further so that some rows are one-hot and some scores reach ~ +-100, against
    fp32 torch autograd over the same bf16 inputs; forward log-sum-exp and output too (the extreme-score check was
    forward-only at T = 200)."""
    lib, L, dev = ctx
    B, H, T, DP = 1, 2, 1654, 160
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B * T, 3, H, DP, generator=g) * 0.7
    x[:, 0] *= 3.5          # q, k: scores ~ N(0, (0.7 * 3.5)^4 * 160 / 160) -> std ~ 6
    x[:, 1] *= 3.5
    x[5, 0] *= 4.0          # a query with scores up to ~ +-100: a one-hot row
    x[900, 1, 0] *= 4.0     # a key (head 0) most queries either lock on to or never see
    x[1653, 0, 1] *= 6.0    # the last query of the ragged tail tile, head 1
    x[:, 2] *= 1.5
    qkv = x.reshape(B * T, 3 * H * DP).to(dev).bfloat16().requires_grad_(True)
    scale = torch.tensor([DP ** -0.5], device=dev)
    # the planes the ViT core uses: the attention output as ONE fp16 plane (2^-12), which the backward's delta = rowsum(dO o O) reads. (With the
    # bf16 output of v1t_attention_forward the one-hot rows of this test - P ~ 1, dP - delta ~ 0 by cancellation - carry delta's rounding
    # error straight into dQ: 6.5e-2 of the tensor's max, GPUTEST of round 6.)
    o = torch.empty(B * T, H * DP, device=dev, dtype=torch.float16)
    lse = torch.empty(B, H, T, device=dev)
    seed, sid = 99, 8
    L.check(lib.v1t_attention_forward_f16o(qkv.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, seed, sid, o.data_ptr(), lse.data_ptr(), L.stream()))
    mask = None
    if p > 0:
        mask = torch.empty(B * H * T, T, device=dev, dtype=torch.uint8)
        L.check(lib.v1t_dropout_mask(seed, sid, p, B * H * T, T, mask.data_ptr(), L.stream()))
    p_eff = float(lib.v1t_attention_dropout_rate(p))
    ref = _attn_ref(qkv, B, H, T, DP, scale.expand(H), mask, p_eff)
    q_, k_ = (qkv.detach().float().view(B, T, 3, H, DP)[:, :, i].permute(0, 2, 1, 3) for i in range(2))
    s_ = (q_ @ k_.transpose(-1, -2)) * float(scale)
    assert 4.0 < float(s_.std()) < 9.0 and float(torch.softmax(s_, -1).max(-1).values.median()) > 0.2  # the regime the test claims
    assert bool(torch.isfinite(o.float()).all()) and bool(torch.isfinite(lse).all())
    check_rel(f"extreme scores p={p}: forward", o.float().cpu(), ref.detach().cpu(), 1.2e-2)
    ref_lse2 = torch.logsumexp(s_, -1) * 1.4426950408889634
    assert float(((lse - ref_lse2).abs() / (1.0 + 4e-3 * ref_lse2.abs())).max()) < 1.0
    dO = (torch.randn(B * T, H * DP, generator=g) * 0.5).to(dev).bfloat16()
    (gq,) = torch.autograd.grad(ref, (qkv,), dO.float())
    delta = torch.empty(B, H, T, device=dev)
    nb = int(lib.v1t_attention_backward_ws_bytes(B, H, T))
    ws = torch.full((nb,), 0xFF, dtype=torch.uint8, device=dev)
    d2 = torch.zeros_like(qkv)
    L.check(lib.v1t_attention_backward_ws_f16o(qkv.data_ptr(), o.data_ptr(), dO.data_ptr(), lse.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, seed, sid,
                                               delta.data_ptr(), d2.data_ptr(), None, ws.data_ptr(), nb, L.stream()))
    assert bool(torch.isfinite(d2.float()).all())
    gq, e = gq.view(B * T, 3, H * DP), d2.float().view(B * T, 3, H * DP)
    errs = []
    for i, nm in enumerate("qkv"):
        try:
            check_grad(f"extreme scores p={p}: d{nm} (dkv2 + dq2, fp16 O plane)", e[:, i].cpu(), gq[:, i].cpu(), 1.2e-2)
        except AssertionError as ex:  # all three tensors are measured (the margins table) before the test fails
            errs.append(str(ex))
    assert not errs, errs


def test_multi_unit_entry_points_beyond_one_table(ctx):
    """The one-launch-per-stage entry points take tables of <= 8 units (24 AdamW ranges) per launch and loop over longer lists: 10 units / 30 ranges
    against their single-unit counterparts (v1t_resize_bilinear + v1t_concat2, v1t_adamw_step), bit for bit."""
    import ctypes as C

    lib, L, dev = ctx
    g = torch.Generator().manual_seed(10)
    # ---- v1t_inputs_multi: 10 units of 1-3 images, resize 18x32 -> 9x16 and behaviour rows cat(3, 2)
    ns = [1, 2, 3, 1, 2, 3, 1, 2, 3, 2]
    imgs = [torch.randn(n, 1, 18, 32, generator=g).to(dev) for n in ns]
    behs = [torch.randn(n, 3, generator=g).to(dev) for n in ns]
    pups = [torch.randn(n, 2, generator=g).to(dev) for n in ns]
    B = sum(ns)
    out, bout = torch.full((B, 1, 9, 16), float("nan"), device=dev), torch.full((B, 5), float("nan"), device=dev)
    VP = C.c_void_p * len(ns)
    L.check(lib.v1t_inputs_multi(VP(*[x.data_ptr() for x in imgs]), VP(*[x.data_ptr() for x in behs]), VP(*[x.data_ptr() for x in pups]), (C.c_int * len(ns))(*ns), len(ns),
                                 1, 18, 32, out.data_ptr(), 9, 16, bout.data_ptr(), 3, 2, L.stream()))
    ref, bref = torch.empty_like(out), torch.empty_like(bout)
    o = 0
    for n, im, be, pu in zip(ns, imgs, behs, pups):
        L.check(lib.v1t_resize_bilinear(im.data_ptr(), n, 18, 32, ref[o:o + n].data_ptr(), 9, 16, L.stream()))
        L.check(lib.v1t_concat2(be.data_ptr(), 3, pu.data_ptr(), 2, n, bref[o:o + n].data_ptr(), 5, L.stream()))
        o += n
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and torch.equal(bout, bref)
    # same sizes in and out: a plain copy
    out2 = torch.empty(B, 1, 18, 32, device=dev)
    L.check(lib.v1t_inputs_multi(VP(*[x.data_ptr() for x in imgs]), None, None, (C.c_int * len(ns))(*ns), len(ns), 1, 18, 32, out2.data_ptr(), 18, 32, None, 0, 0, L.stream()))
    assert torch.equal(out2, torch.cat(imgs))
    # ---- v1t_adamw_multi: 30 ranges of different length / lr / l1 / step against v1t_adamw_step
    sizes = [17 + 97 * i for i in range(30)]
    mk = lambda: [torch.randn(n, generator=g).to(dev) for n in sizes]  # noqa: E731
    p1, g1, m1, v1 = mk(), mk(), mk(), [x.abs() for x in mk()]
    p2, g2, m2, v2 = ([x.clone() for x in t_] for t_ in (p1, g1, m1, v1))
    rs = []
    for i, n in enumerate(sizes):
        lr, l1, step = 1e-3 * (1 + i % 3), (0.0 if i % 2 else 0.01 * i), 1 + i % 5
        rs.append(L.AdamRange(p1[i].data_ptr(), g1[i].data_ptr(), m1[i].data_ptr(), v1[i].data_ptr(), n, lr, l1, step, 0))
        L.check(lib.v1t_adamw_step(p2[i].data_ptr(), g2[i].data_ptr(), m2[i].data_ptr(), v2[i].data_ptr(), n, lr, 0.9, 0.9999, 1e-8, 0.0, step, l1, 1, L.stream()))
    L.check(lib.v1t_adamw_multi((L.AdamRange * len(rs))(*rs), len(rs), 0.9, 0.9999, 1e-8, 0.0, 1, L.stream()))
    torch.cuda.synchronize()
    for a, b in zip(p1 + g1 + m1 + v1, p2 + g2 + m2 + v2):
        assert torch.equal(a, b)
    # ---- v1t_fill_zero: any byte count at a 16-byte aligned address
    buf = torch.full((1000,), 7, dtype=torch.uint8, device=dev)
    L.check(lib.v1t_fill_zero(buf.data_ptr(), 997, L.stream()))
    assert int(buf[:997].sum()) == 0 and int(buf[997:].sum()) == 21


def test_attention_random_shapes(ctx):
    """Shape sweep of the attention entry points (forward, recompute backward, materialised-dS' backward) against fp32 torch: token counts of every
    residue class that matters to the tilings (32-query / 32-key tiles, 64-key stages, 128-key workgroups, 256-query forward workgroups, 512-query
    dQ workgroups), head dims 32-160, 1-3 heads, 1-2 images, with and without dropout. Round 4's two attention bugs sat at shapes no fixed case hit."""
    lib, L, dev = ctx
    rng = torch.Generator().manual_seed(2024)
    Ts = [1, 2, 31, 32, 33, 63, 64, 65, 95, 97, 127, 128, 129, 159, 161, 191, 193, 255, 256, 257, 300, 384, 385, 449, 511, 512, 513, 577, 640, 641, 700, 769, 897, 1025]
    worst = 0.0
    for i, T in enumerate(Ts):
        DP = (160, 160, 128, 96, 64, 32)[i % 6]
        H, B = 1 + i % 3, 1 + (i // 3) % 2
        p = 0.2 if i % 4 == 1 else 0.0
        qkv = (torch.randn(B * T, 3 * H * DP, generator=rng) * 0.7).to(dev).bfloat16()
        dO = (torch.randn(B * T, H * DP, generator=rng) * 0.5).to(dev).bfloat16()
        scale = torch.tensor([DP ** -0.5], device=dev)
        o = torch.full((B * T, H * DP), float("nan"), device=dev, dtype=torch.bfloat16)
        lse = torch.full((B, H, T), float("nan"), device=dev)
        seed, sid = 99 + i, 8
        L.check(lib.v1t_attention_forward(qkv.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, seed, sid, o.data_ptr(), lse.data_ptr(), L.stream()))
        mask, p_eff = None, float(lib.v1t_attention_dropout_rate(p))
        if p > 0:
            mask = torch.empty(B * H * T, T, device=dev, dtype=torch.uint8)
            L.check(lib.v1t_dropout_mask(seed, sid, p, B * H * T, T, mask.data_ptr(), L.stream()))
        q, k, v = (x.detach().clone().requires_grad_(True) for x in qkv.float().view(B, T, 3, H, DP).permute(2, 0, 3, 1, 4))
        a = torch.softmax((q @ k.transpose(-1, -2)) * scale, -1)
        if mask is not None:
            a = a * mask.view(B, H, T, T).float() / (1 - p_eff)
        ref = (a @ v).permute(0, 2, 1, 3).reshape(B * T, H * DP)
        gq, gk, gv = torch.autograd.grad(ref, (q, k, v), dO.float())
        tag = f"T{T} DP{DP} H{H} B{B} p{p}"
        assert bool(torch.isfinite(o.float()).all()) and bool(torch.isfinite(lse).all()), tag
        e = rel_to_max(o.float().cpu(), ref.detach().cpu())
        worst = max(worst, e / 1e-2)
        assert e < 1e-2, f"{tag}: forward {e:.3e}"
        delta = torch.empty(B, H, T, device=dev)
        outs = []
        d1 = torch.full_like(qkv, float("nan"))
        L.check(lib.v1t_attention_backward(qkv.data_ptr(), o.data_ptr(), dO.data_ptr(), lse.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, seed, sid, delta.data_ptr(),
                                           d1.data_ptr(), None, L.stream()))
        outs.append(("recompute", d1))
        nb = int(lib.v1t_attention_backward_ws_bytes(B, H, T))
        ws = torch.full((nb,), 0xFF, dtype=torch.uint8, device=dev)
        d2 = torch.full_like(qkv, float("nan"))
        L.check(lib.v1t_attention_backward_ws(qkv.data_ptr(), o.data_ptr(), dO.data_ptr(), lse.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, seed, sid, delta.data_ptr(),
                                              d2.data_ptr(), None, ws.data_ptr(), nb, L.stream()))
        outs.append(("materialised dS'", d2))
        for nm, d in outs:
            dd = d.float().view(B, T, 3, H, DP).permute(2, 0, 3, 1, 4)
            assert bool(torch.isfinite(dd).all()), f"{tag}: {nm} not finite"
            for j, (r, c) in enumerate(((gq, "q"), (gk, "k"), (gv, "v"))):
                if float(r.abs().max()) == 0.0:
                    continue
                e = rel_to_max(dd[j].cpu(), r.cpu())
                worst = max(worst, e / 1.2e-2)
                assert e < 1.2e-2, f"{tag}: {nm} d{c} {e:.3e}"
    from tests.helpers import record_margin

    record_margin("test_attention_random_shapes: worst error / bound over 34 shapes", worst, 1.0)


def test_fused_poisson_criterion_and_elu1_backward_vs_torch(ctx):
    """The criterion with the reference's call signature (losses.py:141-166, scale_ds :114-119) as one launch and the backward of ELU + 1
    (models/utils.py:109-118) as one launch, against the same arithmetic in plain torch ops (fp64)."""
    from types import SimpleNamespace

    from v1t_amd.losses import PoissonLoss
    from v1t_amd.model import ELU1

    lib, L, dev = ctx
    g = torch.Generator().manual_seed(5)
    B, N = 16, 8000
    u = (torch.randn(B, N, generator=g) * 1.5).to(dev).requires_grad_(True)
    y_true = torch.rand(B, N, generator=g).mul(3.0).to(dev)
    y_true[0, :7] = 0.0  # targets of exactly 0: (0 + eps) log(.)
    ds = {"A": SimpleNamespace(dataset=range(4500))}
    crit = PoissonLoss(SimpleNamespace(ds_scale=1), ds).to(dev)
    y_pred = ELU1().to(dev)(u)
    loss = crit(y_true=y_true, y_pred=y_pred, mouse_id="A", batch_size=B)
    (2.0 * loss).backward()
    u64 = u.detach().double().requires_grad_(True)
    eps = float(torch.finfo(torch.float32).eps)
    yp = torch.nn.functional.elu(u64) + 1 + eps
    ref = math.sqrt(4500 / B) * torch.sum(yp - (y_true.double() + eps) * torch.log(yp))
    (2.0 * ref).backward()
    assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref))
    check_rel("fused_poisson:du", u.grad, u64.grad.float(), 2e-5)
    # no-grad call (validation): no gradient buffer, same value
    with torch.no_grad():
        assert abs(float(crit(y_true=y_true, y_pred=y_pred.detach(), mouse_id="A")) - float(ref)) <= 2e-6 * abs(float(ref))


@pytest.mark.parametrize("switch", ["V1T_MLP_FUSE", "V1T_MLP_BWD_FUSE"])
@pytest.mark.parametrize("images,train", [(3, True), (2, False)])
def test_fused_mlp_forward_equals_two_launches(images, train, switch):
    """The MLP branch forward as ONE launch (gemm.hip mlp_fwd_kernel: LN2 -> FC1 -> GELU -> dropout -> FC2 -> dropout -> + residual; default
    above 256 row tiles, forced here with V1T_MLP_FUSE=2) against ln_gemm + gemm_nt (V1T_MLP_FUSE=0; the switch is read once per process: two
    subprocesses) at the default width (D = 155 -> 160, MLP 488 -> 512), ragged last row tile, dropout masks on (counter-based: identical in
    both runs): core output, and - through the planes the forward saves for the backward (LayerNorm output + statistics, gelu', the fp16
    activation) - the loss gradient of every core parameter. Same operands, same K order of FC2's accumulation.
    V1T_MLP_BWD_FUSE: the same for the backward's mirror (mlp_bwd_kernel: dGELU GEMM + dz GEMM + LN2 backward in one launch against
    gemm_nt<EPI_DGELU> + gemm_lnbwd_kernel): every core gradient."""
    import os
    import subprocess
    import sys

    code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from oracle import v1t_oracle as O
from oracle import weights as W
from tests.helpers import build_native_model
images, train = int(sys.argv[2]), bool(int(sys.argv[3]))
dev = torch.device("cuda:0")
cfg = O.Config(num_blocks=2, emb_dim=155, mlp_dim=488, num_heads=4, patch_stride=2, mouse_ids=("A",), num_neurons={"A": 40},
               p_dropout=0.2 if train else 0.0, t_dropout=0.1 if train else 0.0)
sd = W.make_state_dict(cfg, 7)
model, _ = build_native_model(cfg, sd, dev)
model.train(train)
b = {k: v.to(dev) for k, v in W.make_batch(cfg, "A", images, 7).items()}
model.core.prepare()
model.core._arena.attach_grads()
model.core._arena.grad.zero_()
torch.manual_seed(0)
u = model(inputs=b["image"], mouse_id="A", behaviors=b["behavior"], pupil_centers=b["pupil_center"], activate=False)[0]
(u * torch.linspace(-1, 1, u.numel(), device=dev).view_as(u)).sum().backward()
torch.cuda.synchronize()
torch.save({"u": u.detach().cpu(), "g": model.core._arena.grad.cpu()}, sys.argv[1])
'''
    outs = []
    for fuse in ("0", "2"):
        path = f"/tmp/v1t_mlpfuse_{switch}_{fuse}.pt"
        env = exp_env(**{switch: fuse})
        r = subprocess.run([sys.executable, "-c", code, path, str(images), str(int(train))], env=env, capture_output=True, text=True,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(path))
    a, b = outs
    assert bool(torch.isfinite(b["u"]).all()) and bool(torch.isfinite(b["g"]).all())
    check_rel("test_fused_mlp_forward_equals_two_launches:u", b["u"], a["u"], 2e-5)
    check_rel("test_fused_mlp_forward_equals_two_launches:g", b["g"], a["g"], 2e-5)
