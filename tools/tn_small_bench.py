"""Dev probe (round 6): the four weight-gradient GEMM shapes of a block ALONE on the chip at the row counts of a 16-image launch (the per-mouse loop), a
14-image share and the 112-image step, over the m-chunk (rows per workgroup): v1t_gemm_tn_slab (GEMM + slab reduction), 20 launches, cache flushed by a
1-GB fill between the timings. Prints us per launch, the bytes it must read and the TB/s. usage: python tools/tn_small_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v1t_amd import lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
T = 1654
SHAPES = {"dWqkv (1920 x 160)": (1920, 160), "dW1 (512 x 160)": (512, 160), "dWo (160 x 640)": (160, 640), "dW2 (160 x 512)": (160, 512)}
for images in (16, 14, 112):
    M = images * T
    for name, (NY, NX) in SHAPES.items():
        Y = torch.randn(M, NY, device=dev).bfloat16()
        X = torch.randn(M, NX, device=dev).bfloat16()
        dW = torch.zeros(NY, NX, device=dev)
        tiles = (NY + 127) // 128 if NX == 160 else (NX // 128)
        res = []
        for target in (64, 128, 256, 512, 1024):
            want = max(1, target // tiles)
            mc = max(128, (-(-M // want) + 63) // 64 * 64)
            nb = lib.v1t_gemm_tn_slab_bytes(M, NY, NX, mc)
            if nb <= 0:
                continue
            slab = torch.empty(nb // 4, device=dev)
            for _ in range(3):
                L.check(lib.v1t_gemm_tn_slab(Y.data_ptr(), NY, X.data_ptr(), NX, M, NY, NX, dW.data_ptr(), NX, mc, slab.data_ptr(), nb, L.stream()))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                L.check(lib.v1t_gemm_tn_slab(Y.data_ptr(), NY, X.data_ptr(), NX, M, NY, NX, dW.data_ptr(), NX, mc, slab.data_ptr(), nb, L.stream()))
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            res.append(f"{target}: {us:6.1f} us ({nb / 1e6:5.1f} MB slab)")
        mb = M * (NY + NX) * 2 / 1e6
        best = min(float(r.split(":")[1].split("us")[0]) for r in res)
        print(f"{images:3d} images  {name:20s} {mb:6.1f} MB in  best {best:6.1f} us = {mb / best / 1e6 * 1e6 / 1e6:5.2f} TB/s   | " + "  ".join(res), flush=True)
