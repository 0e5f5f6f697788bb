"""Dev tool: which host-side calls of one training step issue memcpy / fill / ATen kernels (torch.profiler, one step after warm-up)."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

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
for _ in range(3):
    tr.train_step(batches)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step(batches)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    n = e.name
    if n.startswith("aten::") or "Memcpy" in n or "Memset" in n or "hipMemcpy" in n or "hipMemset" in n:
        st = [s for s in (e.stack or []) if "v1t_amd" in s or "bench" in s]
        cnt[(n, st[0] if st else "")] += 1
for (n, s), c in sorted(cnt.items(), key=lambda x: -x[1])[:60]:
    print(f"{c:4d}  {n:40s} {s}")
