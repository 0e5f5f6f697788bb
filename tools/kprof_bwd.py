"""Dev tool: in-kernel s_memtime timeline of the producer / consumer attention backward (attn_bwd_dkv2_kernel). Build the
stamped library with tools/build_kprof.sh; prints per-wave, per-step segment cycles of one workgroup."""
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
dO = (torch.randn(B * T, H * DP, generator=g) * 0.5).to(dev).bfloat16()
scale = torch.tensor([155 ** -0.5], device=dev)
o = torch.empty(B * T, H * DP, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, H, T, device=dev)
dqkv = torch.empty_like(qkv)
delta = torch.empty(B, H, T, device=dev)
nb = int(lib.v1t_attention_backward_ws_bytes(B, H, T))
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
lib.v1t_attention_forward(qkv.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, 1, 8, o.data_ptr(), lse.data_ptr(), L.stream())
for _ in range(3):
    lib.v1t_attention_backward_ws(qkv.data_ptr(), o.data_ptr(), dO.data_ptr(), lse.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, p, 1, 8,
                                  delta.data_ptr(), dqkv.data_ptr(), None, ws.data_ptr(), nb, L.stream())
torch.cuda.synchronize()
raw = C.CDLL(L.LIB_PATH)
NT, NP = 8, 16
buf = (C.c_ulonglong * (8 * NT * NP))()
n = raw.v1t_kprof_read(buf, len(buf))
assert n == len(buf), n
t = np.array(buf, dtype=np.int64).reshape(8, NT, NP)
names = {"producer": ["dma+init", "slots 0-9", "slots 10-19", "vm wait", "barrier"], "consumer": ["dma issue", "hand+store", "mfma issue", "vm wait", "barrier"]}
for w in (0, 1, 4, 5, 7):
    role = "producer" if w < 4 else "consumer"
    print(f"wave {w} ({role})")
    for k in range(NT - 1):
        seg = [t[w, k, i + 1] - t[w, k, i] for i in range(5)]
        print(f"  step {k}: " + "  ".join(f"{nm}={int(v)}" for nm, v in zip(names[role], seg)) + f"  total={int(t[w, k + 1, 0] - t[w, k, 0])}")
