"""Dev tool (round 4): cycles, shader clock and wall time of attn_bwd_dkv2_kernel for several builds (production + ablations, all with -DV1T_KCLK):
does a build that removes cycles get them back as time, or as a lower clock? One process per library (the probe symbol is per library):
  python tools/power_curve.py libv1t_amd_kc.so            -> one line: kernel us (hipEvents, 30 back-to-back backward launches), cycles per workgroup, MHz"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v1t_amd import lib as L  # noqa: E402

name = sys.argv[1]
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), name)
lib = L.load()
raw = C.CDLL(L.LIB_PATH)
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
nb = int(lib.v1t_attention_backward_ws_bytes(B, H, T))
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
lib.v1t_attention_forward(qkv.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, 1, 8, o.data_ptr(), lse.data_ptr(), L.stream())
bw = lambda: lib.v1t_attention_backward_ws(qkv.data_ptr(), o.data_ptr(), dO.data_ptr(), lse.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, 1, 8,  # noqa: E731
                                           delta.data_ptr(), dqkv.data_ptr(), None, ws.data_ptr(), nb, L.stream())
for _ in range(10):
    bw()
torch.cuda.synchronize()
L.check(lib.v1t_profile_enable(2, 64))
cyc, mhz = [], []
for rep in range(30):
    bw()
    if rep % 6 == 5:
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 4)()
        assert raw.v1t_kclk_read(buf) == 0
        cyc.append(buf[0])
        mhz.append(buf[0] / max(buf[1], 1) * 100)
torch.cuda.synchronize()
n, ms = C.c_int(), C.c_double()
L.check(lib.v1t_profile_read(C.byref(n), C.byref(ms)))
print(f"{name:34s} dK/dV kernel {ms.value / max(n.value, 1) * 1e3:8.1f} us   workgroup {sum(cyc) / len(cyc):9.0f} cycles ({sum(cyc) / len(cyc) / 53:6.0f} per step)   "
      f"clock {sum(mhz) / len(mhz):6.0f} MHz", flush=True)
