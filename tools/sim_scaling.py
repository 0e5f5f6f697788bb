"""Dev tool: estimates the N-GPU step time on ONE GPU by running each rank's share of the dealing in turn (no
collectives): max over ranks of the local step time. Usage: python tools/sim_scaling.py [parts ...]
(parts = candidate cut granularities handed to MouseSharding, default its own choice)."""
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
torch.manual_seed(args.seed)
model = v1t_amd.Model(args, ds).to(dev)
batches = {m: make_batch(args, m, 8000, args.batch_size, dev, seed=i) for i, m in enumerate(MICE)}


class NoComm(D.MouseSharding):
    def reduce_core(self, arena):
        pass

    def reduce_mouse(self, mouse_id, arena):
        pass


def step_time(sh, steps=6, warm=2):
    tr = Trainer(args, model, ds, sharding=sh)
    for _ in range(warm):
        tr.train_step(batches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(batches)
    HOST.append((time.perf_counter() - t0) / steps * 1e3)  # the host's own time per step (enqueue only: nothing in a step synchronises)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


HOST = []


if os.environ.get("SIM_ONLY"):  # "world,rank": that rank's share only, 7 steps (for rocprofv3 --kernel-trace --stats / tools/prof_top.py)
    w_, r_ = (int(x) for x in os.environ["SIM_ONLY"].split(","))
    t_ = step_time(NoComm(MICE, r_, w_, args.batch_size, make_groups=False), steps=int(os.environ.get("SIM_STEPS", "5")), warm=2)
    print(f"world {w_} rank {r_}: {t_:.2f} ms (host enqueue time per step {HOST[-1]:.2f} ms)")
    sys.exit(0)
base = step_time(NoComm(MICE, 0, 1, args.batch_size, make_groups=False))
print(f"world 1: {base:.2f} ms")
for world in (2, 4, 8):
    ts = []
    for r in range(world):
        sh = NoComm(MICE, r, world, args.batch_size, make_groups=False)
        ts.append(step_time(sh))
    units = [[(m, None if sl is None else (sl.start, sl.stop)) for m, sl in NoComm(MICE, r, world, args.batch_size, make_groups=False).local_units()] for r in range(world)]
    print(f"world {world}: per-rank ms {[round(t, 2) for t in ts]} -> step {max(ts):.2f} ms, speed-up {base / max(ts):.2f} (+collectives)")
    print("   ", units)
