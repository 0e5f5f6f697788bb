"""Dev tool: error of the predicted responses of the default V1T (golden g2, stress weights) against the reference."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import weights as W
from helpers import build_native_model
dev = torch.device("cuda:0")
golden = np.load(os.path.join(ROOT, "tests", "golden", "g2_default.npz"))
cfg = W.config_c2({"A": 8000})
sd = W.make_state_dict(cfg, 1234)
batch = W.make_batch(cfg, "A", 2, 1234)
model, _ = build_native_model(cfg, sd, dev)
model.train(False)
bd = {k: v.to(dev) for k, v in batch.items()}
with torch.no_grad():
    y = model(inputs=bd["image"], mouse_id="A", behaviors=bd["behavior"], pupil_centers=bd["pupil_center"])[0].cpu().numpy()
ref = golden["g2/y"]
err = np.abs(y - ref)
bound = 1e-3 * np.abs(ref) + 1e-6
print(f"V1T_NOSPLIT={os.environ.get('V1T_NOSPLIT', '0')}: max abs err {err.max():.3e}, worst err/bound {np.max(err / bound):.3f}, mean |y| {np.abs(ref).mean():.3f}")
