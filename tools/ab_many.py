"""Dev tool: forward / backward attention time (112-image launch) of several library builds in ONE process, interleaved rounds.
usage: python tools/ab_many.py lib1.so lib2.so ... (paths relative to v1t_amd/lib or absolute)"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
paths = [p if os.path.isabs(p) else os.path.join(ROOT, "v1t_amd", "lib", p) for p in sys.argv[1:]]
libs = [C.CDLL(p) for p in paths]
dev = torch.device("cuda:0")
B, H, T, DP, p = 112, 4, 1654, 160, 0.2544
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * T, 3 * H * DP, generator=g) * 0.7).to(dev).bfloat16()
dO = (torch.randn(B * T, H * DP, generator=g) * 0.5).to(dev).bfloat16()
scale = torch.tensor([155 ** -0.5], device=dev)
o = torch.empty(B * T, H * DP, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, H, T, device=dev)
dqkv = torch.empty_like(qkv)
delta = torch.empty(B, H, T, device=dev)
P = C.c_void_p
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for lb in libs:
    lb.v1t_attention_backward_ws_bytes.restype = C.c_longlong
nb = max(int(lb.v1t_attention_backward_ws_bytes(B, H, T)) for lb in libs)  # (builds may lay the scratch out differently)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)


def fwd(lb):
    lb.v1t_attention_forward(P(qkv.data_ptr()), B, H, T, DP, P(scale.data_ptr()), 0, 0, C.c_float(p), C.c_uint64(1), C.c_uint32(8), P(o.data_ptr()), P(lse.data_ptr()), st)


def bwd(lb):
    lb.v1t_attention_backward_ws(P(qkv.data_ptr()), P(o.data_ptr()), P(dO.data_ptr()), P(lse.data_ptr()), B, H, T, DP, P(scale.data_ptr()), 0, 0, C.c_float(p),
                                 C.c_uint64(1), C.c_uint32(8), P(delta.data_ptr()), P(dqkv.data_ptr()), None, P(ws.data_ptr()), C.c_longlong(nb), st)


fwd(libs[0])
torch.cuda.synchronize()
o_ref, lse_ref = o.float().clone(), lse.clone()
for i, lb in enumerate(libs[1:], 1):  # same inputs, same dropout stream: the builds must agree
    o.zero_()
    lse.zero_()
    fwd(lb)
    torch.cuda.synchronize()
    print(f"{os.path.basename(paths[i])}: forward vs {os.path.basename(paths[0])}: max |dO| {float((o.float() - o_ref).abs().max()):.3e} (|o| max {float(o_ref.abs().max()):.3f}), "
          f"max |dlse2| {float((lse - lse_ref).abs().max()):.3e}")
res = {(i, k): [] for i in range(len(libs)) for k in ("fwd", "bwd")}
for r in range(5):
    for i, lb in enumerate(libs):
        for k, fn in (("fwd", fwd), ("bwd", bwd)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn(lb)
            e0.record()
            for _ in range(4):
                fn(lb)
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                res[(i, k)].append(e0.elapsed_time(e1) / 4 * 1e3)
for i, lb in enumerate(libs):  # checksum of the backward's output (compare across builds / environment switches)
    dqkv.zero_()
    bwd(lb)
    torch.cuda.synchronize()
    print(f"{os.path.basename(paths[i])}: dqkv checksum {float(dqkv.double().abs().sum()):.9e} nan {int(torch.isnan(dqkv.float()).sum())}")
for i, pth in enumerate(paths):
    f, bw = sorted(res[(i, "fwd")]), sorted(res[(i, "bwd")])
    print(f"{os.path.basename(pth):28s} fwd median {f[len(f) // 2]:8.1f} us   bwd (delta + dK/dV + dQ) median {bw[len(bw) // 2]:8.1f} us")
