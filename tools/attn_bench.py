"""Attention kernel micro-benchmark at the BASELINE shape (B=16, H=4, T=1654, DP=160)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v1t_amd import lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
B, H, T, DP = int(os.environ.get("ATTN_B", "16")), 4, 1654, 160
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * T, 3 * H * DP, generator=g) * 0.7).to(dev).bfloat16()
dO = (torch.randn(B * T, H * DP, generator=g) * 0.5).to(dev).bfloat16()
scale = torch.tensor([155 ** -0.5], device=dev)
o = torch.empty(B * T, H * DP, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, H, T, device=dev)
dqkv = torch.empty_like(qkv)
delta = torch.empty(B, H, T, device=dev)
flops_fwd = 4 * B * H * T * T * 155
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for p in (0.0, 0.2544):
    for cls, name, mult in ((0, "fwd", 1.0), (1, "bwd_dq", 1.5), (2, "bwd_dkv", 2.0)):
        for _ in range(3):
            lib.v1t_attention_forward(qkv.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, 1, 8, o.data_ptr(), lse.data_ptr(), L.stream())
            lib.v1t_attention_backward(qkv.data_ptr(), o.data_ptr(), dO.data_ptr(), lse.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, 1, 8, delta.data_ptr(), dqkv.data_ptr(), None, L.stream())
        torch.cuda.synchronize()
        L.check(lib.v1t_profile_enable(cls, reps + 4))
        for _ in range(reps):
            if cls == 0:
                lib.v1t_attention_forward(qkv.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, 1, 8, o.data_ptr(), lse.data_ptr(), L.stream())
            else:
                lib.v1t_attention_backward(qkv.data_ptr(), o.data_ptr(), dO.data_ptr(), lse.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, 1, 8, delta.data_ptr(), dqkv.data_ptr(), None, L.stream())
        torch.cuda.synchronize()
        n, ms = C.c_int(), C.c_double()
        L.check(lib.v1t_profile_read(C.byref(n), C.byref(ms)))
        if n.value == 0:
            continue
        avg = ms.value / n.value
        print(f"p={p:<6} {name:8s} {avg * 1e3:8.1f} us  executed {mult * flops_fwd / avg / 1e9:7.1f} TFLOP/s ({n.value} launches)", flush=True)
# materialised-dS' backward: dK/dV + store (profile class 2) and the dQ GEMM (class 1), timed in separate passes
nb = int(lib.v1t_attention_backward_ws_bytes(B, H, T))
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
bw = lambda p: lib.v1t_attention_backward_ws(qkv.data_ptr(), o.data_ptr(), dO.data_ptr(), lse.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, 1, 8,
                                             delta.data_ptr(), dqkv.data_ptr(), None, ws.data_ptr(), nb, L.stream())
for p in (0.0, 0.2544):
    tot = 0.0
    for cls, name, mult in ((2, "dkv_store", 2.0), (1, "dq_gemm", 0.5)):
        for _ in range(3):
            bw(p)
        torch.cuda.synchronize()
        L.check(lib.v1t_profile_enable(cls, reps + 4))
        for _ in range(reps):
            bw(p)
        torch.cuda.synchronize()
        n, ms = C.c_int(), C.c_double()
        L.check(lib.v1t_profile_read(C.byref(n), C.byref(ms)))
        avg = ms.value / max(n.value, 1)
        tot += avg
        print(f"p={p:<6} {name:9s} {avg * 1e3:8.1f} us  algorithmic {mult * flops_fwd / avg / 1e9:7.1f} TFLOP/s ({n.value} launches)", flush=True)
    print(f"p={p:<6} bwd (dS')  {tot * 1e3:8.1f} us  algorithmic {2.5 * flops_fwd / tot / 1e9:7.1f} TFLOP/s", flush=True)
lib.v1t_profile_enable(-1, 0)
