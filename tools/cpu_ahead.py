"""How far the host runs ahead of the GPU inside one training step (diagnosis of idle gaps): host timestamps after each
phase of each mouse, without synchronising, then the GPU completion time."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import v1t_amd  # noqa: E402
from v1t_amd.dist import MouseSharding  # noqa: E402
from v1t_amd.losses import elu1_poisson_loss  # noqa: E402
from v1t_amd.synthetic import make_batch, sensorium_config  # noqa: E402
from v1t_amd.trainer import Trainer  # noqa: E402

dev = torch.device("cuda:0")
mice = list("ABCDEFG")
args, ds = sensorium_config({m: 8000 for m in mice})
torch.manual_seed(args.seed)
model = v1t_amd.Model(args, ds).to(dev)
tr = Trainer(args, model, ds, MouseSharding(args.mouse_ids, 0, 1, batch_size=args.batch_size))
batches = {m: make_batch(args, m, 8000, args.batch_size, dev, seed=i) for i, m in enumerate(args.mouse_ids)}
for _ in range(3):
    tr.train_step(batches)
torch.cuda.synchronize()

model.train(True)
core = model.core
marks = []
t0 = time.perf_counter()
mk = lambda name: marks.append((name, (time.perf_counter() - t0) * 1e3))
core.prepare()
mk("prepare")
for m in args.mouse_ids:
    b = batches[m]
    model.mouse_arena(m).attach_grads()
    mk(f"{m} attach")
    u, _, _ = model(inputs=b["image"], mouse_id=m, behaviors=b["behavior"], pupil_centers=b["pupil_center"], activate=False)
    mk(f"{m} fwd")
    loss, _ = elu1_poisson_loss(u, b["response"], 4500.0, 16)
    mk(f"{m} loss")
    loss.backward()
    mk(f"{m} bwd")
t_cpu = (time.perf_counter() - t0) * 1e3
torch.cuda.synchronize()
t_gpu = (time.perf_counter() - t0) * 1e3
prev = 0.0
for n, t_ in marks:
    print(f"{n:12s} +{t_ - prev:7.3f} ms  (at {t_:7.2f})")
    prev = t_
print(f"host done enqueuing at {t_cpu:.2f} ms, GPU done at {t_gpu:.2f} ms")

# ---- whole steps back to back: does anything in train_step block the host until the GPU has drained?
torch.cuda.synchronize()
t0 = time.perf_counter()
hs = []
for _ in range(4):
    tr.train_step(batches)
    hs.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
print("host returned from train_step at", ", ".join(f"{h:.1f}" for h in hs), f"ms; GPU done at {(time.perf_counter() - t0) * 1e3:.1f} ms")
