"""Dev probe (GPU box): core tokens and rollout rows of the default V1T in eval mode with the MLP branch forward as one launch (V1T_MLP_FUSE=2)
against LN2 + FC1 and FC2 as two (V1T_MLP_FUSE=0), same seeded weights, batches of 2 / 24 / 40 images: how many elements differ and by how much
(same operands and K order; the compiler contracts the GELU arithmetic of the two instantiations differently: last-bit differences)."""
import os, sys, subprocess, torch
code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
import v1t_amd
from v1t_amd.synthetic import make_batch, sensorium_config
from v1t_amd.rollout import rollout_rows
dev = torch.device("cuda:0")
torch.manual_seed(0)
args, ds = sensorium_config({"A": 500})
model = v1t_amd.Model(args, ds).to(dev).train(False)
B = int(sys.argv[2])
b = make_batch(args, "A", 500, B, dev, seed=11)
img = model.image_cropper(b["image"], "A", b["behavior"], b["pupil_center"])[0]
core = model.core
outs = []
with torch.no_grad():
    for rep in range(2):
        t = core.forward_tokens(img, "A", b["behavior"], b["pupil_center"]).clone()
        outs.append(t)
    rows = rollout_rows(core, img, b["behavior"], b["pupil_center"], "A")
torch.cuda.synchronize()
print("repeat equal:", bool(torch.equal(outs[0], outs[1])), float((outs[0]-outs[1]).abs().max()))
torch.save({"t": outs[0].cpu(), "rows": rows.cpu()}, sys.argv[1])
'''
res = {}
for B in (2, 24, 40):
    for fuse in ("0", "2"):
        path = f"/tmp/mlp_probe_{B}_{fuse}.pt"
        r = subprocess.run([sys.executable, "-c", code, path, str(B)], env=dict(os.environ, V1T_MLP_FUSE=fuse), capture_output=True, text=True)
        print(B, fuse, r.stdout.strip()[-200:], r.stderr.strip()[-300:] if r.returncode else "")
        res[(B, fuse)] = torch.load(path)
    a, b = res[(B, "0")], res[(B, "2")]
    d = (a["t"] - b["t"]).abs()
    print(f"B={B}: tokens fused vs two launches: max abs diff {float(d.max()):.3e} (max |t| {float(a['t'].abs().max()):.3e}), elements differing {int((d > 0).sum())} of {d.numel()}")
    dr = (a["rows"] - b["rows"]).abs()
    print(f"B={B}: rollout rows: max rel-to-max diff {float(dr.max() / a['rows'].abs().max()):.3e}")
    for i in range(min(B, 4)):
        print("   image", i, "token diff", float(d[i].max()), "rows diff rel", float(dr[i].max() / a["rows"][i].abs().max()))
