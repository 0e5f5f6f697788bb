import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import v1t_amd
from v1t_amd.dist import MouseSharding
from v1t_amd.synthetic import make_batch, sensorium_config
from v1t_amd.trainer import Trainer
dev = torch.device("cuda:0")
args, ds = sensorium_config({"A": 8000})
torch.manual_seed(0)
model = v1t_amd.Model(args, ds).to(dev)
tr = Trainer(args, model, ds, MouseSharding(["A"], 0, 1, batch_size=16))
for B in (1, 2, 4, 8, 16):
    b = {"A": make_batch(args, "A", 8000, B, dev, seed=0)}
    for _ in range(3): tr.train_step(b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): tr.train_step(b)
    torch.cuda.synchronize()
    print(f"B={B:2d}: {(time.perf_counter()-t0)*100:.2f} ms per step")
