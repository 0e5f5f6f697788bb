"""Dev tool: N optimizer steps of the headline configuration on fixed synthetic batches; prints the loss every 20 steps (it must fall and stay finite)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import v1t_amd  # noqa: E402
from v1t_amd.synthetic import make_batch, sensorium_config  # noqa: E402
from v1t_amd.trainer import Trainer  # noqa: E402

dev = torch.device("cuda:0")
MICE = list("ABCDEFG")
args, ds = sensorium_config({m: 8000 for m in MICE})
torch.manual_seed(args.seed)
model = v1t_amd.Model(args, ds).to(dev)
batches = {m: make_batch(args, m, 8000, args.batch_size, dev, seed=i) for i, m in enumerate(MICE)}
tr = Trainer(args, model, ds)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = last = None
for i in range(n):
    o = tr.train_step(batches)
    if i % 20 == 0 or i == n - 1:
        v = float(o["loss"])
        first = v if first is None else first
        last = v
        print(f"step {i:4d} loss {v:.1f}", flush=True)
        assert v == v and abs(v) < 1e12, "loss is not finite"
arena = model.core._arena.data
assert bool(torch.isfinite(arena).all()), "non-finite core parameters"
print(f"loss {first:.1f} -> {last:.1f} ({'falls' if last < first else 'DOES NOT FALL'})")
