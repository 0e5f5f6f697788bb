"""Dev probe (round 6): how much of a 16-image launch of the dK/dV kernel is its quarter-filled last round? The kernel (producer / consumer backward,
v1t_attention_backward_ws_f16o, dropout on) ALONE on the chip at B = 8 .. 18 images (H = 4, T = 1654: 13 key blocks per (image, head) -> 52 B workgroups on 256
CUs), timed per launch by hipEvents on its stream (profile class 2). usage: python tools/dkv2_rounds_probe.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v1t_amd import lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
H, T, DP, p = 4, 1654, 160, 0.2544
g = torch.Generator().manual_seed(0)
for B in (8, 10, 12, 13, 14, 15, 16, 17, 18, 20, 24, 28):
    qkv = (torch.randn(B * T, 3 * H * DP, generator=g) * 0.7).to(dev).bfloat16()
    dO = (torch.randn(B * T, H * DP, generator=g) * 0.5).to(dev).bfloat16()
    scale = torch.tensor([155 ** -0.5], device=dev)
    o = torch.empty(B * T, H * DP, device=dev, dtype=torch.float16)
    lse = torch.empty(B, H, T, device=dev)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B, H, T, device=dev)
    nb = int(lib.v1t_attention_backward_ws_bytes(B, H, T))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    L.check(lib.v1t_attention_forward_f16o(qkv.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, 1, 8, o.data_ptr(), lse.data_ptr(), L.stream()))
    res = {}
    for cls in (2, 1):
        for _ in range(3):
            L.check(lib.v1t_attention_backward_ws_f16o(qkv.data_ptr(), o.data_ptr(), dO.data_ptr(), lse.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, 1, 8,
                                                       delta.data_ptr(), dqkv.data_ptr(), None, ws.data_ptr(), nb, L.stream()))
        torch.cuda.synchronize()
        L.check(lib.v1t_profile_enable(cls, 40))
        for _ in range(20):
            L.check(lib.v1t_attention_backward_ws_f16o(qkv.data_ptr(), o.data_ptr(), dO.data_ptr(), lse.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, 1, 8,
                                                       delta.data_ptr(), dqkv.data_ptr(), None, ws.data_ptr(), nb, L.stream()))
        torch.cuda.synchronize()
        n, ms = C.c_int(), C.c_double()
        L.check(lib.v1t_profile_read(C.byref(n), C.byref(ms)))
        L.check(lib.v1t_profile_enable(-1, 0))
        res[cls] = ms.value / max(n.value, 1) * 1e3
    units = B * H * 13
    print(f"B {B:3d}: dK/dV {units:5d} workgroups = {units / 256:5.2f} rounds: {res[2]:7.1f} us = {res[2] / B:6.2f} us per image | dQ GEMM {B * H * 4:4d} workgroups: {res[1]:6.1f} us = {res[1] / B:5.2f} per image", flush=True)
