"""Dev tool: per-segment cycle sums (KS_MARK probes) of one workgroup of the attention backward dK/dV kernel (KSUM_BWD=1) or forward.
Build the probe library with V1T_BUILD_LIB=libv1t_amd_ksum.so V1T_HIPCC_EXTRA=-DV1T_KSUM python -m v1t_amd.build, then
KSUM_BWD=1 python tools/ksum.py 0.2544 loop,dma,init,slots,vmcnt,barrier,pro,epi"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v1t_amd import lib as L  # noqa: E402

L.LIB_PATH = L.LIB_PATH.replace("libv1t_amd.so", "libv1t_amd_ksum.so")
lib = L.load()
dev = torch.device("cuda:0")
B, H, T, DP = int(os.environ.get("ATTN_B", "112")), 4, 1654, 160
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2544
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * T, 3 * H * DP, generator=g) * 0.7).to(dev).bfloat16()
scale = torch.tensor([155 ** -0.5], device=dev)
o = torch.empty(B * T, H * DP, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, H, T, device=dev)
raw = C.CDLL(L.LIB_PATH)
if os.environ.get("KSUM_BWD"):
    dO = (torch.randn(B * T, H * DP, generator=g) * 0.5).to(dev).bfloat16()
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B, H, T, device=dev)
    nb = int(lib.v1t_attention_backward_ws_bytes(B, H, T))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    lib.v1t_attention_forward(qkv.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, 1, 8, o.data_ptr(), lse.data_ptr(), L.stream())
    torch.cuda.synchronize()
    raw.v1t_ksum_read((C.c_ulonglong * 66)())
for rep in range(6):
    if os.environ.get("KSUM_BWD"):
        lib.v1t_attention_backward_ws(qkv.data_ptr(), o.data_ptr(), dO.data_ptr(), lse.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, 1, 8,
                                      delta.data_ptr(), dqkv.data_ptr(), None, ws.data_ptr(), nb, L.stream())
        continue
    lib.v1t_attention_forward(qkv.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, 1, 8, o.data_ptr(), lse.data_ptr(), L.stream())
torch.cuda.synchronize()
buf = (C.c_ulonglong * 66)()
assert raw.v1t_ksum_read(buf) == 0
names = sys.argv[2].split(",") if len(sys.argv) > 2 else [f"seg{i}" for i in range(8)]
print(f"workgroup: {buf[64]} cycles in {buf[65] / 100:.2f} us -> {buf[64] / max(buf[65], 1) * 100:.0f} MHz")
for w in range(8):
    print(f"wave {w}: " + "  ".join(f"{n}={buf[w * 8 + i]}" for i, n in enumerate(names)))
