"""Validation-path throughput: eval forward of the default V1T (8000 neurons) + metrics over
`trials` trials, (a) with the device-side streaming moments (v1t_amd.metrics.StreamingMetrics) and (b) the reference's
way (train.py:24-25,186: predictions to the host per micro-batch, vstack, metrics on the CPU with torch). Also the
metric kernel's own bandwidth. Not the headline metric (bench.py is)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import v1t_amd  # noqa: E402
from v1t_amd.losses import correlation  # noqa: E402
from v1t_amd.metrics import StreamingMetrics  # noqa: E402
from v1t_amd.synthetic import make_batch, sensorium_config  # noqa: E402

dev = torch.device("cuda:0")
MB = int(sys.argv[1]) if len(sys.argv) > 1 else 64
TRIALS = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
N = 8000
args, ds = sensorium_config({"A": N})
torch.manual_seed(args.seed)
model = v1t_amd.Model(args, ds).to(dev).train(False)
b = make_batch(args, "A", N, MB, dev, seed=0)


def run(streaming: bool):
    sm = StreamingMetrics(N, dev)
    preds, tgts = [], []
    with torch.no_grad():
        for _ in range(TRIALS // MB):
            y, _, _ = model(inputs=b["image"], mouse_id="A", behaviors=b["behavior"], pupil_centers=b["pupil_center"])
            if streaming:
                sm.update(y, b["response"])
            else:
                preds.append(y.cpu())
                tgts.append(b["response"].cpu())
        if streaming:
            return float(sm.correlation(per_neuron=False)), float(sm.msse())
        p, t_ = torch.vstack(preds), torch.vstack(tgts)
        return float(correlation(p, t_, dim=0).mean()), float(torch.square(t_ - p).sum())


for streaming in (True, False):
    run(streaming)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = run(streaming)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{'device streaming' if streaming else 'host vstack (reference way)'}: {TRIALS} trials at micro-batch {MB}: {dt * 1e3:.1f} ms = "
          f"{TRIALS / dt:.0f} images/s (corr {r[0]:.5f}, msse {r[1]:.4e})")

# the accumulate kernel alone on a (4096, 8000) block: 8 B per element algorithmic
p = torch.rand(4096, N, device=dev) + 0.1
t_ = torch.rand(4096, N, device=dev)
sm = StreamingMetrics(N, dev)
sm.update(p, t_)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    sm.update(p, t_)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 10
print(f"metrics_accumulate (4096 x {N}): {us:.1f} us = {2 * p.numel() * 4 / us / 1e3:.0f} GB/s")
