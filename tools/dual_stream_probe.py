"""Dev probe (round 6): do two HALF-batch training steps that run concurrently on two streams finish sooner than one full step?
The step alternates between MFMA-bound kernels (attention: HBM idle) and HBM-bound ones (the GEMM family at K = 160, dQ = dS' . K: matrix
pipe idle); two independent 56-image steps free-running on two streams put the one's GEMMs beside the other's attention. Two models (own
plans / arenas / streams), ranks 0 and 1 of the 2-rank dealing (56 images each), no collectives. Prints: one 112-image step, one 56-image
step alone, both 56-image steps concurrently. usage: python tools/dual_stream_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import v1t_amd  # noqa: E402
from v1t_amd import dist as D  # noqa: E402
from v1t_amd.synthetic import make_batch, sensorium_config  # noqa: E402
from v1t_amd.trainer import Trainer  # noqa: E402

dev = torch.device("cuda:0")
MICE = list("ABCDEFG")
args, ds = sensorium_config({m: 8000 for m in MICE})


class NoComm(D.MouseSharding):
    def reduce_core(self, arena):
        pass

    def reduce_mouse(self, mouse_id, arena):
        pass


def make(rank, world):
    torch.manual_seed(args.seed)
    model = v1t_amd.Model(args, ds).to(dev)
    tr = Trainer(args, model, ds, sharding=NoComm(MICE, rank, world, args.batch_size, make_groups=False))
    return model, tr


batches = {m: make_batch(args, m, 8000, args.batch_size, dev, seed=i) for i, m in enumerate(MICE)}
STEPS, WARM = int(os.environ.get("STEPS", "10")), 3


def timed(fn):
    for _ in range(WARM):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS * 1e3


_, full = make(0, 1)
t112 = timed(lambda: full.train_step(batches))
print(f"one 112-image step: {t112:.2f} ms")
del full
torch.cuda.empty_cache()
(m0, a), (m1, b) = make(0, 2), make(1, 2)
ta = timed(lambda: a.train_step(batches))
tb = timed(lambda: b.train_step(batches))
print(f"56-image steps alone: {ta:.2f} / {tb:.2f} ms (sum {ta + tb:.2f})")
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
for off in (0, 1):
    def both():
        with torch.cuda.stream(s0):
            a.train_step(batches)
        with torch.cuda.stream(s1):
            b.train_step(batches)
    if off:  # start the second stream half a step late: its attention beside the first's GEMMs more often
        with torch.cuda.stream(s1):
            torch.cuda._sleep(int(5e-3 * 2.0e9))
    t2 = timed(both)
    print(f"two 56-image steps on two streams (offset {off}): {t2:.2f} ms per 112 images = {t112 / t2:.3f} x the single 112-image step")
