"""bench.py — training images/s of the MI355X-native V1T hot path on synthetic Sensorium-shaped data.

Metric (BASELINE.json): training images/sec at batch 16 x 7 mice on 1/2/4/8 MI355X.
One "step" = one mouse-batch of 16 images per mouse for 7 mice (fwd + bwd, gradients summed) + one
optimizer step = 112 images (reference train.py:97-111). Default workload = BASELINE config C2: default V1T
(4 blocks, D=155, 4 heads x 155, MLP 488, T=1654 tokens), 7 mice x 8000 neurons, input 1x144x256
resized to 1x36x64 by the cropper stage (inside the timed region), bf16 MFMA / fp32 accumulate,
dropout ON (p=0.0229 / 0.2544, counter-based masks), readout position sampling ON, AdamW + L1.

    python bench.py [--gpus N --steps K --warmup W] [--config c2|c4|c5|c1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

--config (one JSON line each; the default c2 is the headline the driver records):
    c2  BASELINE configs[1] (above)                  c4  configs[3]: Franke-shaped 2x36x64 input, behavior_mode 3, 7 mice x 1121 neurons
    c1  configs[0]: 1-block / 64-d ViT, 1 mouse, 256 neurons, batch 8 (training step, on the GPU path)
    c5  configs[4]: eval forward + attention rollout at batch 256 - ROW-CHAIN rollout (16 MFLOP / image), not the 27 GFLOP
        (T x T) matrix chain of the reference algorithm, of which only row 0 is used (utils/attention_rollout.py:118)

N > 1: per-mouse data parallelism (config C3): mice are sharded over ranks, the shared core's gradient arena is
all-reduced (SUM) over RCCL in per-block buckets behind the backward; total work per step is fixed (112 images)
=> "scaling": "strong". Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant kernel timed
live with hipEvents on its stream; HBM traffic from the tracked PMC summary profiles/r05_pmc_attention.json, which must
describe the launch shape of this run) and, at N = 1, `cpu_baseline` (the CPU oracle timed on the host).
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
PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_attention.json")  # collected by tools/pmc_bench.sh on the GPU box (separate --pmc passes)


def algorithmic_flops(args, n_neurons: int) -> dict:
    """unpadded dims, multiply-add = 2 FLOPs."""
    c, h, w = args.core_input_shape
    P, s = args.patch_size, args.patch_stride
    L = ((h - P) // s + 1) * ((w - P) // s + 1)
    T, D, H, M = L + 1, args.emb_dim, args.num_heads, int(args.mlp_dim)
    blk = 2 * T * D * 3 * H * D + 2 * H * T * T * D * 2 + 2 * T * H * D * D + 4 * T * D * M
    fwd = args.num_blocks * blk + 2 * L * (c * P * P) * D + n_neurons * D * 10
    return {"T": T, "fwd_per_image": fwd, "train_per_image": 3 * fwd, "attn_fwd_per_image_block": 4 * H * T * T * D}


def measured_peak(lib, L) -> dict:
    """The bf16 MFMA rate THIS device sustains (the roofline fraction against the datasheet AND the measured peak):
    `v1t_mfma_peak_probe` runs v_mfma_f32_32x32x16_bf16 back to back on every CU (register operands, random data) in the un-timed
    set-up. MI355X is power-limited under matrix load: the shader clock falls from 2.4 GHz to ~1.5 GHz, so the sustained peak is
    ~1.7 PFLOP/s, not the nominal 2.5 (profiles/r04_lds_peak.txt has the same loop with operands streamed from LDS: ~1.4-1.55)."""
    best = None
    for wps in (1, 2):
        tf, ghz, cpm = C.c_double(), C.c_double(), C.c_double()
        L.check(lib.v1t_mfma_peak_probe(30000 // wps, wps, C.byref(tf), C.byref(ghz), C.byref(cpm), L.stream()), "mfma_peak_probe")
        r = {"tflops": round(tf.value, 1), "clock_ghz": round(ghz.value, 3), "cycles_per_mfma_per_simd": round(cpm.value, 2), "waves_per_simd": wps}
        if best is None or r["tflops"] > best["tflops"]:
            best = r
    return best


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline() -> dict:
    """The CPU oracle (oracle/v1t_oracle.py, verified against the reference in the build container) timed on this host, fp32
    (SURVEY.md 8d / BASELINE.md 3): the C2 training step of ONE mouse at the metric's batch (B = 16: forward + backward, every
    parameter gradient), at two thread counts (32 threads and one per physical core - more threads only add contention at these GEMM
    sizes, so both are reported and `value` is the better one), the C2 eval forward at B = 16 and the C1 training
    step at B = 8. One warm-up + `reps` timed repetitions per leg, bounded so that the whole baseline stays near 30 s."""
    from oracle import v1t_oracle as O
    from oracle import weights as W

    ncpu = os.cpu_count() or 1
    t_start = time.time()

    def timed(fn, reps):
        t0 = time.time()
        fn()
        if time.time() - t0 > 4.0:  # a slow leg (the B = 16 training step on few cores): one timed repetition
            reps = 1
        ts = []
        for _ in range(reps):
            t0 = time.time()
            fn()
            ts.append(time.time() - t0)
        return sorted(ts)[len(ts) // 2]

    legs = {}
    cfg = W.config_c2({"A": 8000})  # C2: default V1T, one mouse x 8000 neurons
    sd = W.make_state_dict(cfg)
    B = 16
    bt, eps = W.make_batch(cfg, "A", B), W.make_eps(cfg, "A", B)

    def c2_train():
        sdd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
        loss, reg, _ = O.total_loss(cfg, sdd, bt, "A", 4500.0, eps=eps, batch_size=16)
        (loss + reg).backward()

    def c2_eval():
        with torch.no_grad():
            O.model_forward(cfg, sd, bt["image"], "A", bt["behavior"], bt["pupil_center"])

    try:
        import psutil

        ncores = psutil.cpu_count(logical=False) or ncpu
    except Exception:  # noqa: BLE001
        ncores = ncpu
    best = None
    for nthr in sorted({min(ncpu, 32), ncores}):
        torch.set_num_threads(nthr)
        if best is not None:
            # the all-cores leg: oversubscribed GEMMs of this size can be an order of magnitude slower than 32 threads (0.13 against
            # 1.5 images/s with 256 threads on the 2 x 64-core round-3 host, 125 s per step). It runs as ONE un-warmed repetition in
            # a child process with a time limit, so that the default bench stays within minutes whatever the host does; a leg that
            # does not finish reports the rate it was slower than.
            limit = 45.0
            code = ("import sys, time, torch; sys.path.insert(0, %r); torch.set_num_threads(%d); "
                    "from oracle import v1t_oracle as O, weights as W; cfg = W.config_c2({'A': 8000}); sd = W.make_state_dict(cfg); "
                    "bt, eps = W.make_batch(cfg, 'A', %d), W.make_eps(cfg, 'A', %d); "
                    "sdd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}; t0 = time.time(); "
                    "l, r, _ = O.total_loss(cfg, sdd, bt, 'A', 4500.0, eps=eps, batch_size=16); (l + r).backward(); print('T', time.time() - t0)") % (ROOT, nthr, B, B)
            import subprocess

            try:
                out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=limit + 40.0).stdout  # + interpreter / torch start-up
                tsec = float([ln for ln in out.splitlines() if ln.startswith("T ")][-1].split()[1])
                rate, note = round(B / tsec, 3), "single un-warmed repetition in a child process"
            except Exception:  # noqa: BLE001  (timeout or failure)
                rate, note = round(B / limit, 3), f"did not finish within {limit:.0f} s: slower than this rate"
        else:
            rate, note = round(B / timed(c2_train, 2), 3), "median of the timed repetitions"
        legs[f"c2_train_{nthr}thr"] = {"images_per_s": rate, "batch": B, "threads": nthr, "timing": note}
        if best is None or rate > best[0]:
            best = (rate, nthr)
    torch.set_num_threads(best[1])
    legs["c2_train"] = {"images_per_s": best[0], "batch": B, "threads": best[1]}
    legs["c2_eval"] = {"images_per_s": round(B / timed(c2_eval, 2), 3), "batch": B, "threads": best[1]}
    cfg1 = W.config_c1()  # C1: 1-block / 64-d ViT, 256 neurons, batch 8
    sd1 = W.make_state_dict(cfg1)
    b1, e1 = W.make_batch(cfg1, "A", 8), W.make_eps(cfg1, "A", 8)

    def c1_train():
        sdd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd1.items()}
        loss, reg, _ = O.total_loss(cfg1, sdd, b1, "A", 4500.0, eps=e1, batch_size=8)
        (loss + reg).backward()

    legs["c1_train"] = {"images_per_s": round(8 / timed(c1_train, 3), 3), "batch": 8, "threads": best[1]}
    return {"value": best[0], "unit": "images/s", "cores": best[1], "kind": "port", "cpu_model": cpu_model(), "legs": legs,
            "sample": f"oracle fp32 (no dropout masks): C2 train step (fwd+bwd) of 1 mouse x 8000 neurons at the metric's batch B={B} with "
                      f"{' and '.join(str(v['threads']) for k, v in legs.items() if k.startswith('c2_train_'))} threads (value = the better), C2 eval "
                      f"forward B={B}, C1 train step B=8; 1 warm-up + 1-3 timed reps each (median; 1 rep when a repetition takes > 4 s), {time.time() - t_start:.0f} s in all, torch "
                      f"{torch.__version__} CPU, host has {ncores} cores / {ncpu} hardware threads"}


def pmc_traffic(kernel: str, images: int, H: int, T: int, DP: int):
    """HBM bytes per launch of `kernel` from the tracked PMC summary (FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes:
    tools/pmc_bench.sh, MI355X_MICROARCH.md); None when the file is missing or was measured on another launch shape
    (the traffic of these kernels is proportional to the images of a launch; heads, tokens and head dim must be equal)."""
    # a missing / mismatching summary costs the line its `traffic` (null, as the contract allows), never the measurement
    if not os.path.exists(PMC_FILE):
        print(f"[bench] {PMC_FILE} is missing (collect it with tools/pmc_bench.sh on the GPU box): roofline.traffic = null", file=sys.stderr, flush=True)
        return None
    d = json.load(open(PMC_FILE))
    sh = d["shape"]
    if (sh["H"], sh["T"], sh["DP"]) != (H, T, DP) or kernel not in d["kernels"]:
        print(f"[bench] {PMC_FILE} was measured on shape {sh} / has no entry for {kernel}; this run launches H={H} T={T} DP={DP}: roofline.traffic = null", file=sys.stderr, flush=True)
        return None
    k = d["kernels"][kernel]
    return (2 * k["fetch_kib"] + k["write_kib"]) * 1024.0 * images / k["images"]


PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E ~8 TB/s nominal (6.3-6.4 TB/s achievable by a streaming read on this chip)
KERNELS = {0: ("attn_fwd", "attn_fwd (S = QK^T, softmax, dropout, P.V)", 1.0),
           1: ("attn_bwd_dq2", "attn_bwd_dq2 (dQ = dS' . K over the materialised dS')", 0.5),
           2: ("attn_bwd_dkv2", "attn_bwd_dkv2 (producer/consumer dK/dV + dS': the products S, dP, dV, dK)", 2.0)}


def run_rollout(a, dev):
    """C5: eval forward (keeping q/k and the log-sum-exp) + head-max of the recomputed probabilities + rollout rows."""
    import v1t_amd
    from v1t_amd import lib as L
    from v1t_amd.rollout import attention_rollouts
    from v1t_amd.synthetic import make_batch, sensorium_config

    B = 256
    args, ds = sensorium_config({"A": a.neurons})
    args.core_input_shape = (1, 36, 64)
    torch.manual_seed(args.seed)
    model = v1t_amd.Model(args, ds).to(dev).train(False)
    b = make_batch(args, "A", a.neurons, B, dev, seed=0)
    lib = L.load()
    with torch.no_grad():
        images, _ = model.image_cropper(b["image"], "A", b["behavior"], b["pupil_center"])
        full = a.rollout == "full"
        for _ in range(max(a.warmup, 1)):
            heat = attention_rollouts(model.core, images, b["behavior"], b["pupil_center"], "A", full_chain=full, keep_scratch=full)
        torch.cuda.synchronize()
        L.check(lib.v1t_profile_enable(7 if full else 0, a.steps * args.num_blocks + 8))
        t0 = time.perf_counter()
        for _ in range(a.steps):
            heat = attention_rollouts(model.core, images, b["behavior"], b["pupil_center"], "A", full_chain=full, keep_scratch=full)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    launches, total_ms = C.c_int(), C.c_double()
    L.check(lib.v1t_profile_read(C.byref(launches), C.byref(total_ms)))
    L.check(lib.v1t_profile_enable(-1, 0))
    fl = algorithmic_flops(args, a.neurons)
    avg_ms = total_ms.value / max(launches.value, 1)
    if full:
        # the reference's algorithm: L products of (T x T) matrices per image (attention_rollout.py:113-117; the first one is with
        # the identity and is executed like the others), 2 T^3 flops each; split-bf16 executes 3 MFMA products per algorithmic one
        T = fl["T"]
        per_launch = 2.0 * T ** 3 * B
        achieved = per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        return {
            "metric": "eval forward + attention rollout images/sec at batch 256 (BASELINE configs[4])", "value": round(B * a.steps / dt, 2), "unit": "images/s",
            "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16+fp16 forward, split-bf16 (bf16x3) rollout products", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]: default V1T eval forward at batch 256 + attention rollout, FULL MATRIX CHAIN as the reference multiplies "
                                   f"it (head-max of the recomputed P per block, then {args.num_blocks} (T x T).(T x T) products per image on the MFMAs, "
                                   f"{2.0 * T ** 3 * args.num_blocks / 1e9:.1f} GFLOP / image), heat-maps {tuple(heat.shape)} included", "global_batch": B},
            "model_tflops_per_s": round((fl["fwd_per_image"] + 2.0 * T ** 3 * args.num_blocks) * B * a.steps / dt / 1e12, 2),
            "roofline": {"kernel": "rollout_matmul (X <- X . A_hat^T, fp32 in / out, 3 bf16 MFMA products per algorithmic product)", "bound": "mfma",
                         "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS / 3.0, "unit": "TFLOP/s", "frac": round(achieved / (PEAK_BF16_TFLOPS / 3.0), 4),
                         "traffic": None, "launches": launches.value, "avg_ms": round(avg_ms, 4), "flops_per_launch": per_launch,
                         "note": "peak = dense bf16 MFMA peak / 3 (split-bf16: three products per algorithmic one)"},
        }
    per_launch = fl["attn_fwd_per_image_block"] * B
    achieved = per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    return {
        "metric": "eval forward + attention rollout images/sec at batch 256 (BASELINE configs[4])", "value": round(B * a.steps / dt, 2), "unit": "images/s",
        "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16+fp16", "data": "synthetic",
        "config": {"workload": "BASELINE configs[4]: default V1T eval forward at batch 256 + attention rollout, ROW-CHAIN form (head-max of the "
                               "recomputed P per block, then 4 vector-matrix products: 16 MFLOP / image; the reference's (T x T) matrix chain, 27 GFLOP / "
                               f"image, is not executed - only its row 0 is used downstream), heat-maps {tuple(heat.shape)} included", "global_batch": B},
        "model_tflops_per_s": round(fl["fwd_per_image"] * B * a.steps / dt / 1e12, 2),
        "roofline": {"kernel": KERNELS[0][1] + ", eval (no dropout)", "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": None if a.no_pmc else pmc_traffic("attn_fwd_eval", B, args.num_heads, fl["T"], model.core.padded_dim),
                     "traffic_source": None if a.no_pmc else "measured offline: " + os.path.relpath(PMC_FILE, ROOT),
                     "launches": launches.value, "avg_ms": round(avg_ms, 4), "flops_per_launch": per_launch},
    }


def count_gpus_without_hip() -> int:
    """GPUs of this node from the KFD topology in sysfs (nodes with simd_count > 0): the launcher must not initialise HIP - a process that has
    is never re-executed on this pool, and `torch.cuda.device_count()` falls through to hipGetDeviceCount on builds without amdsmi. -1: unknown."""
    import glob

    n, seen = 0, False
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(line.split()[:2] for line in open(f) if len(line.split()) >= 2)
        except OSError:
            continue
        seen = True
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    if not seen:
        return -1
    # a restricted visible set (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES): the runtime will offer at most that many
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(a) -> int:
    """`python bench.py --gpus N` without torchrun: N child processes, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as
    torch.distributed.run sets them, rendezvous on 127.0.0.1), rank 0's stdout relayed (its ONE JSON line), the other ranks'
    output to stderr. The parent never initialises the GPU (it counts devices in sysfs; unknown -> the children validate) and never re-executes
    itself: the ranks are fresh children. The children are POLLED: the first non-zero exit terminates the rest (a crashed rank would otherwise
    leave its siblings in the rendezvous or in a collective until the library's timeout) and becomes the exit code; 2 when the node has fewer
    than N GPUs (never a silent single-GPU line). A rank that HANGS is caught twice: every rank runs its own watchdog (`Watchdog`: no progress
    mark for V1T_BENCH_DEADLINE_S seconds -> the rank exits 124, which ends the launch here or under torchrun), and the launcher itself
    terminates all ranks and returns 124 when the whole launch exceeds V1T_BENCH_LAUNCH_DEADLINE_S (default 900 s)."""
    import socket
    import subprocess

    if not a.dry_run:
        n_dev = count_gpus_without_hip()
        if n_dev < 0:
            n_dev = torch.cuda.device_count()  # no KFD topology in sysfs (no driver: the build container): the device list the runtime reports, 0 there
        if n_dev < a.gpus:
            print(f"bench.py: --gpus {a.gpus} but this node has {n_dev} GPU(s)", file=sys.stderr, flush=True)
            return 2
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=None if r == 0 else sys.stderr, stderr=sys.stderr))
    code, live = 0, list(procs)
    t_end = time.time() + float(os.environ.get("V1T_BENCH_LAUNCH_DEADLINE_S", "900"))
    while live:
        if time.time() > t_end:
            print(f"bench.py: launch deadline passed with {len(live)} rank(s) still running: terminating them", file=sys.stderr, flush=True)
            for q_ in live:  # (exact children of this launcher, by handle)
                q_.terminate()
            for q_ in live:
                try:
                    q_.wait(timeout=10)
                except Exception:  # noqa: BLE001
                    q_.kill()
            return code or 124
        for p_ in list(live):
            rc = p_.poll()
            if rc is None:
                continue
            live.remove(p_)
            if rc != 0 and code == 0:
                code = rc
                for q_ in live:  # (exact children of this launcher, by handle)
                    q_.terminate()
        time.sleep(0.05)
    return code


class Watchdog:
    """Per-rank progress deadline. The 8-GPU run of this bench happens once, on hardware nobody can rehearse on: a rank
    that stops making progress - a collective that never completes, a rendezvous that never forms - must END the run instead of sitting in
    the library's timeout. `mark(phase)` records progress; a daemon thread exits the process with code 124 when no mark arrived for
    V1T_BENCH_DEADLINE_S seconds (default 360: above a rendezvous of 8 ranks whose first `import torch` on a fresh box takes minutes and is not
    equally fast on every rank, far below the driver's budget). Under
    torchrun or bench.py's own launcher a rank exiting non-zero terminates its siblings. `on_expire` (rank 0) gets a chance to print the
    failure line first."""

    def __init__(self, rank: int, on_expire=None):
        import threading

        self.rank, self.deadline = rank, float(os.environ.get("V1T_BENCH_DEADLINE_S", "360"))
        self.phase, self.t, self.on_expire, self._stop = "start", time.monotonic(), on_expire, False
        self._th = threading.Thread(target=self._run, daemon=True)
        self._th.start()

    def mark(self, phase: str) -> None:
        self.phase, self.t = phase, time.monotonic()

    def watch_sigterm(self) -> None:
        """Rank 0 only, from the main thread: when the launcher (torchrun or bench.py's own) terminates this rank because ANOTHER rank failed,
        the failure line is still printed. The C-level signal handler writes the signal number to a wake-up socket the moment SIGTERM
        arrives - also while the main thread sits inside a collective or a synchronize - and a daemon thread reads it."""
        import signal
        import socket
        import threading

        r, w = socket.socketpair()
        w.setblocking(False)
        signal.set_wakeup_fd(w.fileno(), warn_on_full_buffer=False)
        signal.signal(signal.SIGTERM, lambda *_: None)
        self._socks = (r, w)

        def run():
            while True:
                b = r.recv(1)
                if b and b[0] == signal.SIGTERM:
                    if self._stop:  # the run is complete (its line is printed): terminated during teardown = success
                        os._exit(0)
                    try:
                        if self.on_expire is not None:
                            self.on_expire(f"terminated by the launcher in phase '{self.phase}' (another rank failed or a deadline passed)")
                    finally:
                        os._exit(143)

        threading.Thread(target=run, daemon=True).start()

    def stop(self) -> None:
        self._stop = True

    def _run(self) -> None:
        while not self._stop:
            time.sleep(0.5)
            if time.monotonic() - self.t > self.deadline and not self._stop:
                msg = f"no progress for {self.deadline:.0f} s in phase '{self.phase}'"
                print(f"[bench] rank {self.rank}: {msg}: exiting 124", file=sys.stderr, flush=True)
                try:
                    if self.on_expire is not None:
                        self.on_expire(msg)
                finally:
                    os._exit(124)


def sources_hash() -> str:
    """sha256 over the attention kernel sources (csrc/attention.hip + attention.h + common.h): stored in the PMC summary by tools/pmc_bench.sh
    and compared here, so that a kernel change that forgot to re-collect the counters shows as `traffic_stale` in the line."""
    import hashlib

    h = hashlib.sha256()
    for f in ("attention.hip", "attention.h", "common.h"):
        with open(os.path.join(ROOT, "v1t_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_stale() -> bool:
    """True when the tracked PMC summary was measured on other attention sources than the ones the LOADED library was built from
    (v1t_amd/build.py records them next to the .so; without that record: the sources in the tree)."""
    try:
        from v1t_amd import lib as L

        built = None
        info = L.LIB_PATH + ".buildinfo.json"
        if os.path.exists(info):
            built = json.load(open(info)).get("attention_sources_sha16")
        return json.load(open(PMC_FILE)).get("sources_sha16") != (built or sources_hash())
    except Exception:  # noqa: BLE001
        return True


def dry_run(a) -> int:
    """Rendezvous check on the CPU (tests/test_bench_launch.py): gloo group over the launcher's environment, one all-reduce."""
    import torch.distributed as dist

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if os.environ.get("V1T_BENCH_FAIL_RANK") == str(rank):  # test hook (tests/test_bench_launch.py): this rank dies before the rendezvous
        return 3
    wd = Watchdog(rank)
    if os.environ.get("V1T_BENCH_HANG_RANK") == str(rank):  # test hook: this rank stops making progress before the rendezvous
        wd.mark("hung on purpose (V1T_BENCH_HANG_RANK)")
        time.sleep(3600)
    wd.mark("rendezvous")
    if world != a.gpus:
        print(f"--gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    seen = torch.ones(1)
    if world > 1:
        import datetime

        dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=float(os.environ.get("V1T_DIST_TIMEOUT_S", "300"))))
        wd.mark("all-reduce")
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        dist.barrier()
        dist.destroy_process_group()
    wd.stop()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_in_group": int(seen.item())}), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--neurons", type=int, default=None)
    ap.add_argument("--config", default="c2", choices=["c1", "c2", "c4", "c5"])
    ap.add_argument("--path", default="native", choices=["native", "module", "module-fused"],
                    help="native: the fused trainer (Trainer.train_step -> _NativeStep; the default and the headline). module: the reference's own loop "
                         "restated over the registry modules - per mouse Model.forward, criterion, (micro/batch) * model.regularizer, .backward(), then "
                         "torch.optim.AdamW.step() over model.get_parameters() (reference train.py:42-116, 216-223): what `train.py --core vit "
                         "--readout gaussian2d` runs after install_into_reference(). module-fused: the same loop with the opt-in v1t_amd.FusedAdamW optimizer")
    ap.add_argument("--rollout", default="row", choices=["row", "full"], help="c5: row-vector chain (default) or the reference's full (T x T) matrix chain")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="roofline.traffic = null instead of reading the tracked PMC summary (used while collecting it)")
    ap.add_argument("--profile-class", type=int, default=2, help="kernel class timed with hipEvents (see include/v1t_amd.h)")
    ap.add_argument("--min-seconds", type=float, default=2.0, help="repeat the K-step timed window until this much timed work has run; the "
                    "median window is reported (0: one window)")
    ap.add_argument("--dry-run", action="store_true", help="launch / rendezvous check only (gloo, no GPU): every rank joins the process group, "
                    "rank 0 prints how many ranks the group's all-reduce saw")
    a = ap.parse_args()

    if int(os.environ.get("LOCAL_RANK", "0")) == 0 and "V1T_LIB" not in os.environ:
        # the library must be the tree's (content hash, v1t_amd/build.py); a no-op when it is, a rebuild (hipcc, before anything touches the
        # GPU) when the snapshot carried a stale one. Other local ranks find it current or fail loudly in lib.load().
        from v1t_amd.build import build as _build

        _build(force=False, verbose=False)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # started without torchrun: become the launcher BEFORE anything touches the GPU (a process that has initialised HIP is
        # never re-executed; the ranks are fresh children)
        raise SystemExit(spawn_ranks(a))
    if a.dry_run:
        raise SystemExit(dry_run(a))

    state = {"line": None, "done": False}  # what rank 0 knows of the line so far: printed with the failure if the run cannot finish

    def fail_line(msg: str) -> None:
        if state.get("done"):  # the success line is out: ONE JSON line per run, whatever happens during teardown
            return
        if int(os.environ.get("RANK", "0")) == 0:
            base = state["line"] or {"metric": "training images/sec at batch 16x7 mice (V1T core vit + gaussian2d readout)", "unit": "images/s", "n_gpus": a.gpus,
                                     "steps": a.steps, "warmup": a.warmup, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "config": {}}
            base = dict(base, value=None, ms_per_step=None, failed=msg)
            base["config"] = dict(base.get("config", {}), exchange=f"failed: {msg}")
            print(json.dumps(base), flush=True)

    wd = Watchdog(int(os.environ.get("RANK", "0")), on_expire=fail_line)
    wd.mark("process group initialisation")
    if int(os.environ.get("RANK", "0")) == 0 and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        wd.watch_sigterm()
    try:
        run_training(a, wd, state)
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001  (an RCCL / HIP error on any rank: the line still prints, with the failure, and rc != 0)
        import traceback

        traceback.print_exc()
        if state.get("done"):  # the measurement completed and its line is printed: a teardown error does not turn it into a failure
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)
        fail_line(f"{type(e).__name__}: {e}")
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(1)  # not sys.exit: a process group whose collective failed can hang in its destructor
    wd.stop()


def run_training(a, wd, state):
    import torch.distributed as dist
    import v1t_amd
    from v1t_amd import lib as L
    from v1t_amd.dist import MouseSharding, describe_rank, init_from_env
    from v1t_amd.synthetic import MOUSE_IDS, default_args, make_batch, make_ds, sensorium_config
    from v1t_amd.trainer import Trainer

    if os.environ.get("V1T_BENCH_RAISE"):  # test hook (tests/test_bench_launch.py): an error inside the run -> the failure line, rc != 0
        raise RuntimeError(os.environ["V1T_BENCH_RAISE"])
    rank, local, world = init_from_env(timeout_s=float(os.environ.get("V1T_DIST_TIMEOUT_S", "300")))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))  # ranks on THIS node (a multi-node launch has world > the node's GPUs)
    if torch.cuda.is_available() and local_world > torch.cuda.device_count() and os.environ.get("V1T_DIST_BACKEND") != "gloo":
        raise SystemExit(f"--gpus {a.gpus}: {local_world} ranks on this node but it has {torch.cuda.device_count()} GPU(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    local = local % torch.cuda.device_count()  # one rank per GPU under the driver; more ranks than GPUs share (dev runs)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if a.config == "c5":
        if world > 1:
            raise SystemExit("config c5 is a single-GPU inference measurement")
        a.neurons = a.neurons or 8000
        line = run_rollout(a, dev)
        if not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
        state["done"] = True
        wd.stop()
        return

    if a.config == "c2":
        n_neur = a.neurons or 8000
        neurons = {m: n_neur for m in MOUSE_IDS}
        args, ds = sensorium_config(neurons)
        args.core_input_shape = (1, 36, 64)
        workload = (f"BASELINE configs[1]: default V1T (4 blocks, D=155, 4 heads, MLP 488, T=1654) + Gaussian2d, 7 mice x {n_neur} neurons, input "
                    "1x144x256 -> 36x64, batch 16 per mouse, dropout+sampling on, AdamW+L1")
        metric = "training images/sec at batch 16x7 mice (V1T core vit + gaussian2d readout)"
    elif a.config == "c4":
        n_neur = a.neurons or 1121
        neurons = {m: n_neur for m in MOUSE_IDS}
        args, ds = sensorium_config(neurons, input_shape=(2, 36, 64), ds_name="franke2022", behavior_mode=3)
        args.core_input_shape = (2, 36, 64)
        workload = ("BASELINE configs[3]: Franke2022-shaped 2-channel input 2x36x64 (no resize), behavior_mode 3 (BehaviorMLP token injection), default V1T, "
                    f"7 mice x {n_neur} neurons, batch 16 per mouse, dropout+sampling on, AdamW+L1")
        metric = "training images/sec at batch 16x7 mice, Franke-shaped input (BASELINE configs[3])"
    else:  # c1
        n_neur = a.neurons or 256
        neurons = {"A": n_neur}
        args = default_args(input_shape=(1, 36, 64), resize_image=0, num_blocks=1, emb_dim=64, mlp_dim=128, num_heads=4, batch_size=8)
        args.output_shapes = {"A": (n_neur,)}
        args.mouse_ids = ["A"]
        args.core_input_shape = (1, 36, 64)
        ds = make_ds(neurons, seed=args.seed)
        workload = f"BASELINE configs[0]: 1-block / 64-d ViT (4 heads, MLP 128), 1 mouse (36x64 gray, {n_neur} neurons), batch 8, training step"
        metric = "training images/sec at batch 8, 1 mouse (BASELINE configs[0] on the GPU path)"
    torch.manual_seed(args.seed)  # identical initial core on every rank
    model = v1t_amd.Model(args, ds).to(dev)
    torch.manual_seed(args.seed + 7919 * (rank + 1))  # after the (identical) initialisation: per-rank draws of eps / DropPath
    wd.mark("mouse sharding / group creation")
    sharding = MouseSharding(args.mouse_ids, rank=rank, world=world, batch_size=args.batch_size)
    if world > 1:  # before the first collective: who runs what, where (a hung rank is identifiable from the log)
        print(f"[bench] {describe_rank(sharding, dev)} (pid {os.getpid()}, {torch.cuda.get_device_name(dev)})", file=sys.stderr, flush=True)
    trainer = Trainer(args, model, ds, sharding=sharding)
    if a.path != "native" and world > 1:
        raise SystemExit("--path module is the reference's single-device loop (train.py:42-116); the data-parallel step is --path native")
    batches = {m: make_batch(args, m, neurons[m], args.batch_size, dev, seed=i) for i, m in enumerate(args.mouse_ids) if m in sharding.local_mice()}

    lib = L.load()
    ranks_seen, backend = 1, "none"
    state["line"] = {"metric": metric, "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "higher_is_better": True, "scaling": "strong",
                     "vs_baseline": None, "dtype": "bf16+fp16", "data": "synthetic",
                     "config": {"workload": workload, "global_batch": sharding.images_per_step(), "parallelism": f"mouse-dp{world}"}}
    if world > 1:
        wd.mark("first collective (all-reduce of ones)")
        cnt = torch.ones(1, device=dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)  # how many ranks the collective library really joined (goes into the JSON line)
        ranks_seen, backend = int(cnt.item()), str(dist.get_backend())
        # self-check of the exchange before anything is timed: the overlapped, bucketed all-reduce the trainer uses must give
        # the sums of one blocking all-reduce over the same buffer; otherwise fall back to the blocking form (and say so)
        core = model.core
        core.prepare()
        core._arena.attach_grads()
        n = core._arena.param_floats
        wd.mark("exchange self-check")
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
    step_path = {"v": None}
    if a.path == "native":
        def one_step():
            o = trainer.train_step(batches)
            step_path["v"] = trainer.last_step_path
            return o["loss"]
    else:
        # The reference's loop restated (train.py:42-116): per mouse Model.forward -> criterion -> (micro / batch) * model.regularizer ->
        # backward (gradients accumulate over the mice), then optimizer.step() + zero_grad(). micro-batch = batch (on 288 GB the
        # reference's compute_micro_batch_size reaches the batch size, utils/utils.py:431-464). The optimizer is torch.optim.AdamW over
        # model.get_parameters(core_lr) exactly as train.py:216-223 builds it ("module"), or the opt-in fused optimizer with the same
        # interface over the flat arenas ("module-fused", v1t_amd.FusedAdamW.for_model).
        from v1t_amd.losses import PoissonLoss

        criterion = PoissonLoss(args, ds)
        core_lr = args.lr if getattr(args, "core_lr", None) is None else args.core_lr
        if a.path == "module":
            optimizer = torch.optim.AdamW(params=model.get_parameters(core_lr=core_lr), lr=args.lr, betas=(args.adam_beta1, args.adam_beta2), eps=args.adam_eps, weight_decay=0)
        else:
            optimizer = v1t_amd.FusedAdamW.for_model(model, lr=args.lr, core_lr=core_lr, betas=(args.adam_beta1, args.adam_beta2), eps=args.adam_eps)
        model.train(True)
        optimizer.zero_grad()
        step_path["v"] = "per-mouse"

        def one_step():
            total = None
            for m in args.mouse_ids:
                b = batches[m]
                bs = b["image"].size(0)
                y_pred, _, _ = model(inputs=b["image"], mouse_id=m, behaviors=b["behavior"], pupil_centers=b["pupil_center"])
                loss = criterion(y_true=b["response"], y_pred=y_pred, mouse_id=m, batch_size=bs)
                reg = (b["response"].size(0) / bs) * model.regularizer(m)
                tl = loss + reg
                tl.backward()
                total = loss.detach() if total is None else total + loss.detach()
            optimizer.step()
            optimizer.zero_grad()
            return total

    wd.mark("warm-up steps")
    for _ in range(a.warmup):
        one_step()
        wd.mark("warm-up steps")
    torch.cuda.synchronize()
    units_per_step = len(sharding.local_units())
    n_local_launches = 16 * a.steps * max(units_per_step, 1) * args.num_blocks  # up to 15 windows + margin (per-mouse launches on the module path)
    L.check(lib.v1t_profile_enable(a.profile_class, n_local_launches + 8))
    host_s = []

    def window():
        """EXACTLY a.steps steps between barrier + synchronize on both sides; the maximum over ranks."""
        wd.mark("timed window")
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            o = one_step()
        host_s.append(time.perf_counter() - t0)  # the host's enqueue time of the K steps (it runs ahead of the GPU unless it is the limit)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt_ = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt_], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
        return dt_, o

    # A window of K steps is ~0.5 s at the default K: shorter than the sampling period of an outside observer (the driver's
    # gpu_busy sampler saw 0 % in round 2). The window is therefore repeated until >= 2 s of timed GPU work have run (every
    # window is exactly K steps, identically bracketed) and the MEDIAN window is reported; all windows are listed.
    dt, out = window()
    n_win = 1 if a.min_seconds <= 0 else max(1, min(15, int(-(-a.min_seconds // dt))))
    wins = [dt]
    for _ in range(n_win - 1):
        dt_i, out = window()
        wins.append(dt_i)
    dt = sorted(wins)[len(wins) // 2]
    launches, total_ms = C.c_int(), C.c_double()
    L.check(lib.v1t_profile_read(C.byref(launches), C.byref(total_ms)))
    L.check(lib.v1t_profile_enable(-1, 0))
    loss = float(out)
    # the second-largest kernel family is HBM-bound: the dQ GEMM over the materialised dS' (class 1). Timed live the same way over a few EXTRA,
    # un-timed steps behind the windows (the event timing serves one kernel class at a time) -> `roofline_hbm` of the line
    hbm_launches, hbm_ms = C.c_int(), C.c_double()
    if world == 1 and a.profile_class == 2:
        wd.mark("roofline_hbm steps")
        L.check(lib.v1t_profile_enable(1, 4 * 3 * max(units_per_step, 1) * args.num_blocks + 8))
        for _ in range(3):
            one_step()
        torch.cuda.synchronize()
        L.check(lib.v1t_profile_read(C.byref(hbm_launches), C.byref(hbm_ms)))
        L.check(lib.v1t_profile_enable(-1, 0))
    wd.mark("measured-peak probe / report")
    # un-timed, BEHIND the timed windows (75 ms of back-to-back MFMAs in front of them would hand the first window a pre-heated chip):
    # the chip is as warm as the step left it, so the clock the probe reports is the loaded one
    peak_m = measured_peak(lib, L) if rank == 0 else None
    torch.cuda.synchronize()

    if rank == 0:
        images = sharding.images_per_step() * a.steps
        fl = algorithmic_flops(args, n_neur)
        # dominant kernel (class 2): the producer / consumer dK/dV kernel = 4 of the 5 algorithmic products of the flash backward
        # (S, dP, dV, dK) = 2.0 x the forward attention FLOPs per launch; the fifth (dQ = dS' . K) is the HBM-bound GEMM (class 1)
        key, label, mult = KERNELS.get(a.profile_class, (None, str(a.profile_class), 0.0))
        # images per launch: the trainer runs the shared core over all local mouse-batches at once (one launch per block)
        units = sharding.local_units()
        per_rank = sum(args.batch_size if sl is None else (sl.stop - sl.start) for _, sl in units)
        imgs_launch = min(per_rank, trainer.core_group * args.batch_size) if (trainer.batch_core and len(units) > 1) else args.batch_size
        if step_path["v"] == "per-mouse":
            imgs_launch = args.batch_size  # one core pass per mouse-batch
        per_launch = mult * fl["attn_fwd_per_image_block"] * imgs_launch
        DPad = (args.emb_dim + 31) // 32 * 32
        traffic = pmc_traffic(key, imgs_launch, args.num_heads, fl["T"], DPad) if (key and DPad >= 128 and not a.no_pmc) else None
        avg_ms = total_ms.value / max(launches.value, 1)
        achieved = per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        line = {
            "metric": metric,
            "value": round(images / dt, 2),
            "unit": "images/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3),
            "windows": {"n": len(wins), "ms_per_step_each": [round(w / a.steps * 1e3, 3) for w in wins], "reported": "median window of exactly K steps",
                        "host_enqueue_ms_per_step": round(min(host_s) / a.steps * 1e3, 3)},
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "bf16+fp16",
            "data": "synthetic",
            "config": {"workload": workload, "global_batch": sharding.images_per_step(), "parallelism": f"mouse-dp{world}", "ranks_in_group": ranks_seen, "backend": backend,
                       "exchange": ("bucketed async all-reduce behind per-block events" if trainer.overlap else "blocking all-reduce") if world > 1 else "none",
                       # which step ran: "native" = the fused trainer's fixed C-ABI sequence over all local mice; "batched-autograd" = all local mice
                       # through the shared core in one pass under torch autograd; "per-mouse" = the reference's loop over mice (train.py:97-111)
                       "step_path": step_path["v"], "path": a.path,
                       "optimizer": {"native": "v1t_amd.FusedAdamW (L1 folded in)", "module": "torch.optim.AdamW (train.py:216-223)", "module-fused": "v1t_amd.FusedAdamW.for_model (opt-in)"}[a.path]},
            "loss": loss,
            "model_tflops_per_s": round(fl["train_per_image"] * images / dt / 1e12, 2),
            "model_frac_of_bf16_peak": round(fl["train_per_image"] * images / dt / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
            "roofline": {"kernel": label, "bound": "mfma", "achieved": round(achieved, 2),
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": None if traffic is None else "measured offline: " + os.path.relpath(PMC_FILE, ROOT),
                         # the counters were collected on the kernel sources whose hash the summary carries; True = the library loaded now was built
                         # from other attention sources (re-run tools/pmc_bench.sh)
                         "traffic_stale": None if traffic is None else pmc_stale(),
                         "launches": launches.value, "avg_ms": round(avg_ms, 4), "flops_per_launch": per_launch,
                         "peak_measured": peak_m["tflops"], "frac_of_measured": round(achieved / peak_m["tflops"], 4), "peak_measured_clock_ghz": peak_m["clock_ghz"],
                         "peak_measured_how": f"v1t_mfma_peak_probe: v_mfma_f32_32x32x16_bf16 back to back on all CUs, register operands, random data, "
                                              f"{peak_m['waves_per_simd']} wave(s) per SIMD, {peak_m['cycles_per_mfma_per_simd']} cycles per MFMA and SIMD; nominal clock 2.4 GHz"},
        }
        # the backward hands a block's weight-gradient GEMMs (HBM-bound) to a second stream, where they run beside the NEXT block's dK/dV
        # kernel (compute-bound): its live duration then includes the sharing (alone: V1T_DW_SIDE=0, ~12 % shorter per launch, the step slower)
        if a.profile_class == 2 and int(lib.v1t_vit_backward_second_stream(model.core._plan, int(imgs_launch))):
            line["roofline"]["shares_gpu_with"] = ("gemm_tn2 / tn_reduce_multi of the previous block on a second stream (weight gradients; "
                                                   "V1T_DW_SIDE=0 runs everything on one stream: this kernel ~12 % shorter, the step ~1.3 % slower)")
        line["model_frac_of_measured_peak"] = round(fl["train_per_image"] * images / dt / 1e12 / (peak_m["tflops"] * world), 4)
        if hbm_launches.value > 0 and DPad == 160:
            # algorithmic bytes of one dQ-GEMM launch: dS' read once (bf16, T padded to 128 both ways) + k read + dQ written
            TPQ = (fl["T"] + 127) // 128 * 128
            alg = imgs_launch * args.num_heads * (TPQ * TPQ * 2 + 2 * fl["T"] * DPad * 2)
            h_ms = hbm_ms.value / hbm_launches.value
            h_traffic = pmc_traffic("attn_bwd_dq2", imgs_launch, args.num_heads, fl["T"], DPad) if not a.no_pmc else None
            line["roofline_hbm"] = {"kernel": KERNELS[1][1], "bound": "hbm", "achieved": round(alg / (h_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                    "frac": round(alg / (h_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "traffic": h_traffic, "launches": hbm_launches.value, "avg_ms": round(h_ms, 4),
                                    "bytes_per_launch": alg, "how": "hipEvents around every launch over 3 extra un-timed steps behind the timed windows; every other "
                                    "kernel of the step: profiles/r06_roofline_table.md (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE, duration alone and live)"}
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    # The measurement is complete (rank 0: printed). From here on nothing may print a second line or turn the run into a failure: disarm the
    # watchdog / failure paths first, then leave WITHOUT the process group's destructor - it can hang or raise when a sibling is already gone
    # (ADVICE r05), and a completed 8-GPU measurement must not be recorded as failed because of its teardown.
    state["done"] = True
    wd.stop()
    if world > 1:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
