"""Config C5: attention-rollout throughput of the default V1T at batch 256 (eval forward that keeps q/k and
the log-sum-exp, head-max of the recomputed probabilities per block, rollout rows as a vector-matrix chain, min-max
normalisation and resize). Prints images/s; not the headline metric (bench.py is)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import v1t_amd  # noqa: E402
from v1t_amd.rollout import attention_rollouts  # noqa: E402
from v1t_amd.synthetic import make_batch, sensorium_config  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
args, ds = sensorium_config({"A": 8000})
torch.manual_seed(args.seed)
model = v1t_amd.Model(args, ds).to(dev).train(False)
b = make_batch(args, "A", 8000, B, dev, seed=0)
with torch.no_grad():
    images, _ = model.image_cropper(b["image"], "A", b["behavior"], b["pupil_center"])
    for _ in range(2):
        heat = attention_rollouts(model.core, images, b["behavior"], b["pupil_center"], "A")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        heat = attention_rollouts(model.core, images, b["behavior"], b["pupil_center"], "A")
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"rollout: batch {B}, {dt * 1e3:.1f} ms per batch = {B / dt:.0f} images/s, heat-map {tuple(heat.shape)}, finite {bool(torch.isfinite(heat).all())}")
