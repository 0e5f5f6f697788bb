"""Dev tool: where the time of the full-chain rollout at batch 256 goes (phases timed with events)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import v1t_amd
from v1t_amd import lib as L
from v1t_amd.synthetic import make_batch, sensorium_config

dev = torch.device("cuda:0")
B = int(os.environ.get("RB", "256"))
args, ds = sensorium_config({"A": 8000})
args.core_input_shape = (1, 36, 64)
model = v1t_amd.Model(args, ds).to(dev).train(False)
b = make_batch(args, "A", 8000, B, dev, seed=0)
core = model.core
lib = L.load()
with torch.no_grad():
    images, _ = model.image_cropper(b["image"], "A", b["behavior"], b["pupil_center"])
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tokens = core.forward_tokens(images, "A", b["behavior"], b["pupil_center"], keep_workspace=True)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        T = core.num_tokens; TP = (T + 3) // 4 * 4
        cfg = core._cfg; H, DP = cfg.num_heads, core.padded_dim
        A = torch.empty((B, T, TP), dtype=torch.float32, device=dev)
        rowsum = torch.empty((B, T), dtype=torch.float32, device=dev)
        X = [torch.empty((B, T, TP), dtype=torch.float32, device=dev) for _ in range(2)]
        torch.cuda.synchronize(); t2 = time.perf_counter()
        cur = None
        th = tm = 0.0
        for k in range(cfg.num_blocks):
            qkv = core.workspace_tensor("qkv", k)[:B * T * 3 * H * DP * 2]
            lse2 = core.workspace_tensor("lse2", k)[:B * H * T * 4]
            scale = core.transformer.blocks[k]["mha"].scale
            torch.cuda.synchronize(); a0 = time.perf_counter()
            L.check(lib.v1t_rollout_headmax(qkv.data_ptr(), lse2.data_ptr(), B, H, T, DP, scale.data_ptr(), 0, 0, A.data_ptr(), TP, rowsum.data_ptr(), L.stream()))
            torch.cuda.synchronize(); a1 = time.perf_counter()
            out = X[k & 1]
            L.check(lib.v1t_rollout_matmul(A.data_ptr(), rowsum.data_ptr(), L.ptr(cur), out.data_ptr(), B, T, TP, L.stream()))
            torch.cuda.synchronize(); a2 = time.perf_counter()
            cur = out
            th += a1 - a0; tm += a2 - a1
        rows = cur[:, 1:T, 0].contiguous()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        print(f"forward {1e3*(t1-t0):.1f} ms  alloc {1e3*(t2-t1):.1f}  headmax {1e3*th:.1f}  matmul {1e3*tm:.1f}  total chain {1e3*(t3-t2):.1f}", flush=True)
        del A, X, rowsum, cur, out
