"""bench.py — training images/s of the MI355X-native V1T hot path on synthetic Sensorium-shaped data.

Metric (BASELINE.json): training images/sec at batch 16 x 7 mice on 1/2/4/8 MI355X.
One "step" = one mouse-batch of 16 images per mouse for 7 mice (fwd + bwd, gradients summed) + one
optimizer step = 112 images (reference train.py:97-111). Workload = BASELINE config C2: default V1T
(4 blocks, D=155, 4 heads x 155, MLP 488, T=1654 tokens), 7 mice x 8000 neurons, input 1x144x256
resized to 1x36x64 by the cropper stage (inside the timed region), bf16 MFMA / fp32 accumulate,
dropout ON (p=0.0229 / 0.2544, counter-based masks), readout position sampling ON, AdamW + L1.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: per-mouse data parallelism (config C3): mice are sharded over ranks, the shared core's gradient
arena is all-reduced (SUM) over RCCL; total work per step is fixed (112 images) => "scaling": "strong".
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant kernel timed
live with hipEvents on its stream) and, at N = 1, `cpu_baseline` (the CPU oracle timed on the host).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def algorithmic_flops(args, n_neurons: int) -> dict:
    """unpadded dims, multiply-add = 2 FLOPs."""
    c, h, w = 1, 36, 64
    P, s = args.patch_size, args.patch_stride
    L = ((h - P) // s + 1) * ((w - P) // s + 1)
    T, D, H, M = L + 1, args.emb_dim, args.num_heads, int(args.mlp_dim)
    blk = 2 * T * D * 3 * H * D + 2 * H * T * T * D * 2 + 2 * T * H * D * D + 4 * T * D * M
    fwd = args.num_blocks * blk + 2 * L * (c * P * P) * D + n_neurons * D * 10
    return {"T": T, "fwd_per_image": fwd, "train_per_image": 3 * fwd, "attn_fwd_per_image_block": 4 * H * T * T * D}


def cpu_baseline(seconds_budget: float = 30.0) -> dict:
    """The CPU oracle (oracle/v1t_oracle.py, verified against the reference in the build container)
    timed on this host: default-V1T single-mouse train step (fwd + bwd, fp32) on a bounded sample."""
    from oracle import v1t_oracle as O
    from oracle import weights as W

    cores = min(os.cpu_count() or 1, 32)  # more threads than this only adds contention at these GEMM sizes
    torch.set_num_threads(cores)
    cfg = W.config_c2({"A": 8000})
    sd = W.make_state_dict(cfg)
    B = 2
    batch = W.make_batch(cfg, "A", B)
    eps = W.make_eps(cfg, "A", B)

    def step():
        sdd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
        loss, reg, _ = O.total_loss(cfg, sdd, batch, "A", 4500.0, eps=eps, batch_size=16)
        (loss + reg).backward()

    step()  # warm-up
    t0 = time.time()
    n = 0
    while True:
        step()
        n += 1
        if time.time() - t0 > seconds_budget * 0.5 or n >= 8:  # about 10-15 s of CPU work
            break
    dt = (time.time() - t0) / n
    return {"value": round(B / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32 train step (fwd+bwd, no dropout masks), default V1T, 1 mouse x 8000 neurons, B={B}, {n} reps after 1 warm-up, torch {torch.__version__} CPU, {cores} threads"}


# (FETCH_SIZE, WRITE_SIZE) in KiB per 112-image launch (all 7 mouse-batches of the default shape in one core pass), dropout
# on: profiles/r01_pmc_attention_fetch_write.txt, tools/pmc_bench.sh (separate --pmc passes; FETCH_SIZE x2 on gfx950)
PMC_IMAGES = 112
PMC_KIB = {"attn_fwd": (422222.9, 522369.0), "attn_bwd_fused": (166139.7 * 7, 104235.2 * 7), "attn_bwd_dkv_store": (623565.6, 2895070.2),
           "attn_bwd_dq_gemm": (1368689.1, 234988.2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--neurons", type=int, default=8000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-class", type=int, default=2, help="kernel class timed with hipEvents (see include/v1t_amd.h)")
    a = ap.parse_args()

    import torch.distributed as dist
    import v1t_amd
    from v1t_amd import lib as L
    from v1t_amd.dist import MouseSharding, init_from_env
    from v1t_amd.synthetic import MOUSE_IDS, make_batch, sensorium_config
    from v1t_amd.trainer import Trainer

    rank, local, world = init_from_env()
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    local = local % torch.cuda.device_count()  # one rank per GPU under the driver; more ranks than GPUs share (dev runs)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    neurons = {m: a.neurons for m in MOUSE_IDS}
    args, ds = sensorium_config(neurons)
    torch.manual_seed(args.seed)  # identical initial core on every rank
    model = v1t_amd.Model(args, ds).to(dev)
    torch.manual_seed(args.seed + 7919 * (rank + 1))  # after the (identical) initialisation: per-rank draws of eps / DropPath
    sharding = MouseSharding(args.mouse_ids, rank=rank, world=world, batch_size=args.batch_size)
    trainer = Trainer(args, model, ds, sharding=sharding)
    batches = {m: make_batch(args, m, neurons[m], args.batch_size, dev, seed=i) for i, m in enumerate(args.mouse_ids) if m in sharding.local_mice()}

    lib = L.load()
    if world > 1:
        # self-check of the exchange before anything is timed: the overlapped, bucketed all-reduce the trainer uses must give
        # the sums of one blocking all-reduce over the same buffer; otherwise fall back to the blocking form (and say so)
        core = model.core
        core.prepare()
        core._arena.attach_grads()
        n = core._arena.param_floats
        probe = torch.arange(n, device=dev, dtype=torch.float32).remainder_(97.0).mul_(rank + 1.0)
        ref = probe.clone()
        dist.all_reduce(ref, op=dist.ReduceOp.SUM)
        if trainer.overlap:
            try:
                core._arena.grad[:n].copy_(probe)
                sharding.attach_block_events(core)
                sharding.wait_all(sharding.reduce_core_overlapped(core))
                torch.cuda.synchronize()
                ok = bool(torch.allclose(core._arena.grad[:n], ref, rtol=1e-5, atol=1e-3))
            except Exception as e:  # noqa: BLE001
                print(f"[bench] rank {rank}: overlapped exchange raised {e!r}", file=sys.stderr, flush=True)
                ok = False
            flag = torch.tensor([1.0 if ok else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if float(flag.item()) < 1.0:
                if rank == 0:
                    print("[bench] overlapped exchange failed its self-check: using the blocking all-reduce", file=sys.stderr, flush=True)
                trainer.overlap = False
        core._arena.grad.zero_()
    for _ in range(a.warmup):
        trainer.train_step(batches)
    torch.cuda.synchronize()
    n_local_launches = a.steps * len(sharding.local_units()) * args.num_blocks
    L.check(lib.v1t_profile_enable(a.profile_class, n_local_launches + 8))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = trainer.train_step(batches)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    launches, total_ms = C.c_int(), C.c_double()
    L.check(lib.v1t_profile_read(C.byref(launches), C.byref(total_ms)))
    L.check(lib.v1t_profile_enable(-1, 0))
    loss = float(out["loss"])

    if rank == 0:
        images = sharding.images_per_step() * a.steps
        fl = algorithmic_flops(args, a.neurons)
        # dominant kernel: attn_bwd_dkv_store (dK/dV of flash backward + the materialised dS'): 4 of the 5 algorithmic
        # products of the backward (S, dP, dV, dK) = 2.0 x forward attention FLOPs per launch; the fifth (dQ = dS' . K) is
        # the separate HBM-bound GEMM attn_bwd_dq_gemm (class 1, 0.5 x). With V1T_ATTN_BWD_DS=0 class 2 is the fused
        # recompute kernel instead (all 5 products, 2.5 x) - the label below follows the switch.
        fused = os.environ.get("V1T_ATTN_BWD_DS", "1") == "0"
        names = {0: "attn_fwd", 1: "attn_bwd_dq_gemm (dQ = dS' . K)", 2: "attn_bwd_fused (dQ + dK/dV bodies)" if fused else "attn_bwd_dkv_store (dK/dV + dS')"}
        mult = {0: 1.0, 1: 0.5, 2: 2.5 if fused else 2.0}[a.profile_class] if a.profile_class in (0, 1, 2) else 0.0
        # images per launch: the trainer runs the shared core over all local mouse-batches at once (one launch per block)
        units = sharding.local_units()
        per_rank = sum(args.batch_size if sl is None else (sl.stop - sl.start) for _, sl in units)
        imgs_launch = min(per_rank, trainer.core_group * args.batch_size) if (trainer.batch_core and len(units) > 1) else args.batch_size
        per_launch = mult * fl["attn_fwd_per_image_block"] * imgs_launch
        # HBM bytes per launch of that kernel from the PMC counters (FETCH_SIZE x 2 + WRITE_SIZE, KiB, separate --pmc
        # passes: profiles/r01_pmc_attention_fetch_write.txt, tools/pmc_attn.sh); recorded, not collected live, and
        # only valid for the shape it was measured on (16 images x 4 heads x 1654 tokens x 160 padded head dim)
        default_shape = args.batch_size == 16 and a.neurons == 8000
        traffic = {0: PMC_KIB["attn_fwd"], 1: PMC_KIB["attn_bwd_dq_gemm"], 2: PMC_KIB["attn_bwd_fused" if fused else "attn_bwd_dkv_store"]}.get(a.profile_class)
        # (measured on 112-image launches; the kernels' traffic is proportional to the images of a launch)
        traffic = (2 * traffic[0] + traffic[1]) * 1024 * imgs_launch / PMC_IMAGES if (default_shape and traffic) else None
        avg_ms = total_ms.value / max(launches.value, 1)
        achieved = per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        line = {
            "metric": "training images/sec at batch 16x7 mice (V1T core vit + gaussian2d readout)",
            "value": round(images / dt, 2),
            "unit": "images/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "bf16+fp16",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: default V1T (4 blocks, D=155, 4 heads, MLP 488, T=1654) + Gaussian2d, 7 mice x "
                                   f"{a.neurons} neurons, input 1x144x256 -> 36x64, batch 16 per mouse, dropout+sampling on, AdamW+L1",
                       "global_batch": sharding.images_per_step(), "parallelism": f"mouse-dp{world}",
                       "exchange": ("bucketed async all-reduce behind per-block events" if trainer.overlap else "blocking all-reduce") if world > 1 else "none"},
            "loss": loss,
            "model_tflops_per_s": round(fl["train_per_image"] * images / dt / 1e12, 2),
            "model_frac_of_bf16_peak": round(fl["train_per_image"] * images / dt / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
            "roofline": {"kernel": names.get(a.profile_class, str(a.profile_class)), "bound": "mfma", "achieved": round(achieved, 2),
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                         "launches": launches.value, "avg_ms": round(avg_ms, 4), "flops_per_launch": per_launch},
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
