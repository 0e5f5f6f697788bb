"""Dev tool: how much of the 1e-3 parity bound the predicted responses use, golden g2 (default V1T, 8000 neurons) and g2b / g1."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import weights as W
from tests.helpers import build_native_model
golden = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g1_g2.npz")) if os.path.exists(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g1_g2.npz")) else None
import glob
files = {os.path.basename(f): np.load(f) for f in glob.glob(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "*.npz"))}
def find(key):
    for n, z in files.items():
        if key in z.files: return z[key]
    raise KeyError(key)
dev = torch.device("cuda:0")
for name, cfg in (("g1", W.config_c1()), ("g2", W.config_c2({"A": 8000})), ("g2b", W.config_c4())):
    sd = W.make_state_dict(cfg, 1234)
    batch = W.make_batch(cfg, "A", 2, 1234)
    model, _ = build_native_model(cfg, sd, dev)
    model.train(False)
    bd = {k: v.to(dev) for k, v in batch.items()}
    with torch.no_grad():
        y = model(inputs=bd["image"], mouse_id="A", behaviors=bd["behavior"], pupil_centers=bd["pupil_center"])[0].cpu().numpy()
    ref = find(f"{name}/y")
    err = np.abs(y - ref); bound = 1e-6 + 1e-3 * np.abs(ref)
    print(f"{name}: worst {float((err / bound).max()):.3f} of the bound, max abs err {err.max():.2e}")
