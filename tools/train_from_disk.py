"""End-to-end run from the on-disk Sensorium layout: recordings written by oracle/fake_sensorium.py at
the real image shape -> get_training_ds (packed store in HBM) -> Trainer steps -> validate / evaluate. Prints the loader
rate next to the reference's way of producing a batch (per-trial .npy reads + host standardisation) and the step rate.
Usage (GPU box): python tools/train_from_disk.py [trials_per_mouse] [neurons]"""
import os
import sys
import tempfile
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import v1t_amd  # noqa: E402
from oracle import fake_sensorium as FS  # noqa: E402  (test infrastructure: only writes the synthetic recording)
from v1t_amd.data import get_training_ds  # noqa: E402
from v1t_amd.dist import MouseSharding  # noqa: E402
from v1t_amd.evaluate import evaluate, validate  # noqa: E402
from v1t_amd.losses import PoissonLoss  # noqa: E402
from v1t_amd.synthetic import default_args  # noqa: E402
from v1t_amd.trainer import Trainer  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
neurons = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
dev = torch.device("cuda:0")
FS.DIRS[("sensorium", "B")] = "static22846-10-16-GrayImageNet-94c6ff995dac583098847cfecd43e7b6"
mice = ["A", "B"]
with tempfile.TemporaryDirectory() as root:
    t0 = time.perf_counter()
    for i, m in enumerate(mice):
        FS.write_fake_mouse(root, "sensorium", m, seed=i, trials=trials, image_shape=(1, 144, 256), neurons=neurons)
    print(f"wrote {len(mice)} x {trials} trials in {time.perf_counter() - t0:.1f} s")
    args = default_args(input_shape=(1, 144, 256), resize_image=1, batch_size=16)
    args.ds_name, args.mouse_ids, args.limit_data, args.gray_scale, args.micro_batch_size = "sensorium", mice, None, False, 16
    t0 = time.perf_counter()
    train_ds, val_ds, test_ds = get_training_ds(args, data_dir=root, mouse_ids=mice, batch_size=16, device=dev)
    it = {m: iter(train_ds[m]) for m in mice}
    first = {m: next(it[m]) for m in mice}
    torch.cuda.synchronize()
    print(f"packed stores built in {time.perf_counter() - t0:.1f} s; image store dtype {v1t_amd.data.MouseStore.get(train_ds['A'].dataset.mouse_dir, dev).data['image'].dtype}")
    # loader rate: device gather + standardise vs the reference's per-trial host path (same arithmetic, MiceDataset.__getitem__)
    n = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        for b in train_ds["A"]:
            n += len(b["image"])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"device loader: {n / dt:.0f} images/s")
    ds = train_ds["A"].dataset
    t0 = time.perf_counter()
    k = min(len(ds), 160)
    for i in range(k):
        ds[i]
    print(f"host per-trial path (4 .npy reads + numpy transforms, 1 core, page cache warm): {k / (time.perf_counter() - t0):.0f} images/s")
    torch.manual_seed(0)
    model = v1t_amd.Model(args, train_ds).to(dev)
    tr = Trainer(args, model, train_ds, MouseSharding(mice, 0, 1, batch_size=16))
    steps, losses = 0, []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for epoch in range(2):
        for batches in zip(*[train_ds[m] for m in mice]):
            bd = {m: b for m, b in zip(mice, batches)}
            if min(len(b["image"]) for b in bd.values()) < 16:
                continue
            losses.append(tr.train_step(bd)["loss"])
            steps += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ls = torch.stack(losses).cpu().numpy()
    print(f"{steps} steps from disk-format data: {steps * 16 * len(mice) / dt:.0f} images/s incl. loading; loss {ls[0]:.4e} -> {ls[-1]:.4e}")
    crit = PoissonLoss(args, val_ds).to(dev)
    print("validate:", {k: round(v, 4) for k, v in validate(args, val_ds, model, crit).items()})
    print("evaluate:", {k: round(v, 4) for k, v in evaluate(args, test_ds, model).items()})
