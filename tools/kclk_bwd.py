"""Dev tool: shader clock held by the attention backward (attn_bwd_dkv2_kernel): cycles and 100 MHz ticks of one workgroup.
Build: hipcc ... -DV1T_KCLK (tools/build_kclk.sh) -> libv1t_amd_kclk.so"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v1t_amd import lib as L  # noqa: E402

L.LIB_PATH = L.LIB_PATH.replace("libv1t_amd.so", "libv1t_amd_kclk.so")
lib = L.load()
dev = torch.device("cuda:0")
B, H, T, DP = 112, 4, 1654, 160
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2544
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
raw = C.CDLL(L.LIB_PATH)
for rep in range(20):
    lib.v1t_attention_backward_ws(qkv.data_ptr(), o.data_ptr(), dO.data_ptr(), lse.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, 1, 8,
                                  delta.data_ptr(), dqkv.data_ptr(), None, ws.data_ptr(), nb, L.stream())
    torch.cuda.synchronize()
    if rep % 5 == 4:
        buf = (C.c_ulonglong * 4)()
        assert raw.v1t_kclk_read(buf) == 0
        cyc, ticks = buf[0], buf[1]
        print(f"workgroup: {cyc} cycles in {ticks / 100:.2f} us -> {cyc / max(ticks, 1) * 100:.0f} MHz; per step {cyc / 53:.0f} cycles")
