"""Dev tool: N optimizer steps of the headline configuration through the REFERENCE's loop restated (per mouse Model.forward, criterion,
(micro / batch) * model.regularizer, backward; optimizer.step(), zero_grad(): train.py:42-116) over the registry modules, with torch.optim.AdamW
(argv[2] = "fused": v1t_amd.FusedAdamW.for_model). Prints the summed loss every 20 steps: it must fall like tools/train_sanity.py's (same
batches, same initial weights; the dropout / noise streams differ)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import v1t_amd  # noqa: E402
from v1t_amd.losses import PoissonLoss  # noqa: E402
from v1t_amd.synthetic import make_batch, sensorium_config  # noqa: E402

dev = torch.device("cuda:0")
MICE = list("ABCDEFG")
args, ds = sensorium_config({m: 8000 for m in MICE})
torch.manual_seed(args.seed)
model = v1t_amd.Model(args, ds).to(dev)
batches = {m: make_batch(args, m, 8000, args.batch_size, dev, seed=i) for i, m in enumerate(MICE)}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
fused = len(sys.argv) > 2 and sys.argv[2] == "fused"
core_lr = args.lr if getattr(args, "core_lr", None) is None else args.core_lr
if fused:
    opt = v1t_amd.FusedAdamW.for_model(model, lr=args.lr, core_lr=core_lr, betas=(args.adam_beta1, args.adam_beta2), eps=args.adam_eps)
else:
    opt = torch.optim.AdamW(params=model.get_parameters(core_lr=core_lr), lr=args.lr, betas=(args.adam_beta1, args.adam_beta2), eps=args.adam_eps, weight_decay=0)
crit = PoissonLoss(args, ds)
model.train(True)
opt.zero_grad()
first = last = None
for i in range(n):
    total = None
    for m in MICE:
        b = batches[m]
        bs = b["image"].size(0)
        y, _, _ = model(inputs=b["image"], mouse_id=m, behaviors=b["behavior"], pupil_centers=b["pupil_center"])
        loss = crit(y_true=b["response"], y_pred=y, mouse_id=m, batch_size=bs)
        (loss + (b["response"].size(0) / bs) * model.regularizer(m)).backward()
        total = loss.detach() if total is None else total + loss.detach()
    opt.step()
    opt.zero_grad()
    if i % 20 == 0 or i == n - 1:
        v = float(total)
        first = v if first is None else first
        last = v
        print(f"step {i:4d} loss {v:.1f}", flush=True)
        assert v == v and abs(v) < 1e12, "loss is not finite"
assert bool(torch.isfinite(model.core._arena.data).all()), "non-finite core parameters"
print(f"{'FusedAdamW.for_model' if fused else 'torch.optim.AdamW'}: loss {first:.1f} -> {last:.1f} ({'falls' if last < first else 'DOES NOT FALL'})")
