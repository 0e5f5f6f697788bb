"""Dev tool: in-kernel s_memtime timeline of the attention forward kernel (build with V1T_KPROF=1:
`HIPCC_EXTRA=-DV1T_KPROF python -m v1t_amd.build --force`). Prints per-wave, per-tile segment cycles."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v1t_amd import lib as L  # noqa: E402

L.LIB_PATH = L.LIB_PATH.replace("libv1t_amd.so", "libv1t_amd_kprof.so")
lib = L.load()
dev = torch.device("cuda:0")
B, H, T, DP = 16, 4, 1654, 160
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2544
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * T, 3 * H * DP, generator=g) * 0.7).to(dev).bfloat16()
scale = torch.tensor([155 ** -0.5], device=dev)
o = torch.empty(B * T, H * DP, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, H, T, device=dev)
for _ in range(3):
    lib.v1t_attention_forward(qkv.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, 1, 8, o.data_ptr(), lse.data_ptr(), L.stream())
torch.cuda.synchronize()
raw = C.CDLL(L.LIB_PATH)
NT, NP = 8, 16
buf = (C.c_ulonglong * (8 * NT * NP))()
n = raw.v1t_kprof_read(buf, len(buf))
assert n == len(buf), n
t = np.array(buf, dtype=np.int64).reshape(8, NT, NP)
names = ["loop", "K+S chain", "max", "resc+dma", "softmax", "PV issue", "dma wait", "barrier"]
for w in (0, 1, 4, 5):
    print(f"wave {w}")
    for k in range(NT - 1):
        seg = [t[w, k, 0] - t[w, k, 7]] + [t[w, k, i + 1] - t[w, k, i] for i in range(6)] + [t[w, k + 1, 7] - t[w, k, 6]]
        print("     pieces: " + " ".join(str(int(t[w, k, i + 1] - t[w, k, i])) for i in range(8, 14)))
        print(f"  tile {k}: " + "  ".join(f"{nm}={int(v)}" for nm, v in zip(names, seg)) + f"  total={int(t[w, k + 1, 0] - t[w, k, 0])}")
