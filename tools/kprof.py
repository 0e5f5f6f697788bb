"""Dev tool: in-kernel s_memtime timeline of the attention forward kernel. Build the probe library with
    V1T_BUILD_LIB=libv1t_amd_kprof.so V1T_HIPCC_EXTRA=-DV1T_KPROF python -m v1t_amd.build
and run `V1T_LIB=libv1t_amd_kprof.so python tools/kprof.py [p]` on the GPU box: mean cycles per segment of a 32-key tile for one
workgroup's waves (tiles KP_T0 .. KP_T0 + KP_NT), even tiles (which issue the next stage's LDS-DMA) and odd tiles (followed by the
stage barrier) separately. Stamps (attention.hip): 0 tile start, 1 K reads + S chain issued, 2 S complete + row max, 3 rescale
check + DMA issue, 4 softmax / dropout / packing, 5 P.V issued; the rest up to the next tile's stamp 0 is the wait for the chain,
the stage's vmcnt wait and the barrier."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v1t_amd import lib as L  # noqa: E402

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
names = ["K+S issue", "S done+max", "resc+dma", "softmax", "PV issue", "to next"]
print(f"p = {p}; cycles per segment, mean over tiles of each parity (KP_T0 = 8 is even)")
for w in range(8):
    for par in (0, 1):
        rows = []
        for k in range(par, NT - 1, 2):
            seg = [t[w, k, i + 1] - t[w, k, i] for i in range(5)] + [t[w, k + 1, 0] - t[w, k, 5]]
            rows.append(seg + [t[w, k + 1, 0] - t[w, k, 0]])
        m = np.mean(np.array(rows, dtype=np.float64), axis=0)
        print(f"wave {w} {'even' if par == 0 else 'odd '} tiles: " + "  ".join(f"{nm}={v:6.0f}" for nm, v in zip(names, m[:-1])) + f"   total={m[-1]:6.0f}")
