"""Readout kernel micro-benchmark at the BASELINE shape (B=16, C=155, 29x57 cells, N=8000)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v1t_amd import lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
B, C, H, W, N, DP = 16, 155, 29, 57, 8000, 160
g = torch.Generator().manual_seed(0)
z = torch.randn(B, H * W + 1, DP, generator=g).to(dev)
grid = (torch.rand(B, N, 2, generator=g) * 2 - 1).to(dev)
feat = torch.randn(N, DP, generator=g).to(dev)
gout = torch.randn(B, N, generator=g).to(dev)
dz = torch.zeros_like(z)
dgrid = torch.empty_like(grid)
dfeat = torch.zeros_like(feat)
dbias = torch.zeros(N, device=dev)


nb = lib.v1t_gaussian2d_backward_ws_bytes(B, H, W, N)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
use_ws = len(sys.argv) < 2 or sys.argv[1] != "atomics"


def run():
    L.check(lib.v1t_gaussian2d_backward_ws(z.data_ptr() + 4 * DP, (H * W + 1) * DP, DP, B, C, H, W, N, grid.data_ptr(), feat.data_ptr(), DP,
                                           gout.data_ptr(), dz.data_ptr() + 4 * DP, (H * W + 1) * DP, DP, dgrid.data_ptr(), dfeat.data_ptr(),
                                           dbias.data_ptr(), ws.data_ptr() if use_ws else None, nb if use_ws else 0, L.stream()))


out = torch.empty(B, N, device=dev)
bias = torch.zeros(N, device=dev)


def fwd():
    L.check(lib.v1t_gaussian2d_forward(z.data_ptr() + 4 * DP, (H * W + 1) * DP, DP, B, C, H, W, N, grid.data_ptr(), feat.data_ptr(), DP, bias.data_ptr(), out.data_ptr(), L.stream()))


def part(p):
    def f():
        L.check(lib.v1t_gaussian2d_backward_parts(z.data_ptr() + 4 * DP, (H * W + 1) * DP, DP, B, C, H, W, N, grid.data_ptr(), feat.data_ptr(), DP, gout.data_ptr(),
                                                  dz.data_ptr() + 4 * DP, (H * W + 1) * DP, DP, dgrid.data_ptr(), dfeat.data_ptr(), dbias.data_ptr(), ws.data_ptr(), nb, p, L.stream()))
    return f


def timeit(name, fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / reps * 1e6:.1f} us per call")


timeit("readout forward", fwd)
if use_ws:
    timeit("backward: tap sort", part(1))
    timeit("backward: parameter gradients", part(2))
    timeit("backward: dz gather", part(4))
for _ in range(5):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    run()
torch.cuda.synchronize()
print(f"readout backward: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us per call")
