"""profiles/rNN_pmc_all.json (tools/pmc_all.sh: rocprofv3 FETCH_SIZE / WRITE_SIZE passes + kernel-trace durations of one bench.py step)
-> profiles/rNN_roofline_table.md: for every kernel above 0.3 % of the step, the measured HBM-side traffic (FETCH_SIZE x 2 + WRITE_SIZE:
the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md), the achieved TB/s against 8 TB/s nominal and against the streaming
ceilings of this chip (profiles/r04_hbm_stream.txt: read 6.4, write 4.7, copy 5.2 TB/s), achieved TFLOP/s, and traffic / algorithmic bytes.

Algorithmic bytes = every tensor the kernel must read or write ONCE, listed per kernel below (BASELINE configs[1], one 112-image launch:
R = 112 x 1654 token rows, padded dims DP 160 / HDP 640 / MP 512). usage: python tools/roofline_table.py [r06] [tag]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
tag = sys.argv[2] if len(sys.argv) > 2 else ""
d = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_all{tag}.json")))
B, H, T, D, DP, M = 112, 4, 1654, 155, 160, 488
R = B * T
N_NEUR, MICE = 8000, 7
x32, p160, qkv, o640, h512 = R * 160 * 4, R * 160 * 2, R * 1920 * 2, R * 640 * 2, R * 512 * 2
TPQ = (T + 127) // 128 * 128
dS = B * H * TPQ * TPQ * 2
attn_f = 4 * B * H * T * T * D  # forward attention FLOPs (2 products)
readout_alg = MICE * (16 * 1653 * 155 * 4 + 155 * N_NEUR * 4 + 16 * N_NEUR * 4 * 4)  # z + features + grid / bias / out per mouse
mice_floats, core_floats = MICE * 1_280_000, 2_465_200
# kernel -> (what, bound, algorithmic bytes, algorithmic FLOPs, the tensors counted)
MODEL = {
    "attn_fwd_kernel<160, true, false, false>": ("QK^T, softmax, P-dropout, PV (vit.py:253-265)", "mfma", qkv + o640, attn_f, "q k v; O fp16, lse"),
    "attn_bwd_dkv2_kernel<160, true>": ("S, dP, dV, dK + materialised dS'", "mfma", qkv + o640 + 2 * o640, 2 * attn_f, "q k v dO; dK dV (dS' NOT counted: it is the design's own round trip)"),
    "attn_bwd_dq2_kernel<160, true>": ("dQ = dS' . K", "hbm", dS + 2 * o640, attn_f // 2, "dS' k; dQ"),
    "ln_gemm_kernel<160, 0, 4, 2, 2>": ("LN1 -> QKV (+ BehaviorMLP injection)", "hbm", x32 + qkv + 2 * p160 + x32, 2 * R * D * 3 * H * D, "x fp32; qkv, z1 bf16 + fp16 planes, x + beta fp32"),
    "mlp_fwd_kernel<160, 1>": ("LN2 -> FC1 -> GELU -> FC2 -> + residual", "hbm", x32 + p160 + 2 * h512 + x32, 4 * R * D * M, "x fp32; z2, gelu' plane, hact fp16, x out fp32"),
    "gemm_nt_kernel<5, 2, 4, 32, true, false, 0>": ("proj + bias + dropout + residual", "hbm", o640 + 2 * x32, 2 * R * H * D * D, "O fp16, x fp32; x out"),
    "gemm_nt_kernel<5, 0, 4, 32, false, true, 0>": ("dO = dA . Wo + row constants -keep rowsum(dO o O), -lse", "hbm", p160 + 2 * o640, 2 * R * H * D * D, "dA bf16, O; dO"),
    "gemm_nt_kernel<4, 4, 4, 32, false, false, 0>": ("dGELU: dh = (dY . W2) * gelu' * mask", "hbm", p160 + 2 * h512, 2 * R * D * M, "dY, gelu' plane; dh"),
    "gemm_lnbwd_kernel<5, true, 0>": ("dX GEMM + LayerNorm backward + residual (4 x dFC1 form, 3 x dQKV form per step)", "hbm",
                                      (4 * (h512 + 3 * x32 + p160) + 3 * (qkv + 3 * x32 + p160)) // 7, (4 * 2 * R * D * M + 3 * 2 * R * D * 3 * H * D) // 7, "dY, x, G in; G out, dy of the next branch"),
    "gemm_lnbwd_kernel<5, false, 0>": ("dQKV form of block 0 (no next branch)", "hbm", qkv + 3 * x32, 2 * R * D * 3 * H * D, "dqkv, x, G in; G out"),
    "gemm_tn2_kernel<1, 5, false, 2>": ("dWqkv = dqkv^T z1, dW1 = dh^T z2 (one of each per block)", "hbm", (qkv + p160 + h512 + p160) // 2, (2 * R * D * 3 * H * D + 2 * R * D * M) // 2, "dY, X (both read once)"),
    "gemm_tn2_kernel<5, 1, true, 2>": ("dWo = dA^T O (320 workgroups: double buffer, two per CU)", "hbm", p160 + o640, 2 * R * D * H * D, "dA, O fp16 plane"),
    "gemm_tn2_kernel<5, 1, true, 3>": ("dW2 = dY^T hact (256 workgroups: 3-stage ring)", "hbm", p160 + h512, 2 * R * D * M, "dY, activation fp16 plane"),
    "readout_fwd_multi_kernel<3>": ("bilinear taps . features + bias, 7 mice (gaussian2d.py:270-276)", "hbm", readout_alg, MICE * 16 * N_NEUR * D * 10, "z, features, grid, bias, out (SURVEY 8d)"),
    "readout_bwd_multi_kernel<3>": ("d features / d bias / d grid", "hbm", readout_alg + MICE * 155 * N_NEUR * 4, MICE * 16 * N_NEUR * D * 12, "z, features, gout; d features"),
    "readout_dz_gather_multi_kernel<3>": ("dz by sorted taps", "hbm", x32 + MICE * (155 * N_NEUR * 4 + 16 * N_NEUR * 4 * 6), MICE * 16 * N_NEUR * D * 8, "features, taps; dz fp32"),
    "adamw_multi_kernel": ("AdamW + L1, 7 mouse arenas", "hbm", mice_floats * 4 * 8, mice_floats * 12, "p g m v read; p g m v written"),
    "adamw_kernel": ("AdamW + L1, core arena", "hbm", core_floats * 4 * 8, core_floats * 12, "p g m v read; p g m v written"),
    "patch_unfold_kernel": ("Unfold -> bf16 + fp16 patch planes", "hbm", B * 2304 * 4 + 2 * R * 128 * 2, 0, "images; U planes"),
    "gemm_nt_kernel<5, 5, 4, 32, true, false, 0>": ("patch projection + bias + pos + cls + dropout", "hbm", R * 128 * 2 + x32, 2 * R * 64 * D, "U fp16; x0"),
    "patch_bwd_pos_cast_kernel": ("d pos / d cls + bf16 gradient", "hbm", x32 + p160, 0, "d x0; bf16 copy"),
    "drop_cast_kernel<5>": ("dropout backward + cast of the core output gradient", "hbm", x32 + p160, 0, "gout; dy"),
    "readout_sort_multi_kernel": ("counting sort of the taps", "hbm", MICE * 16 * N_NEUR * 4 * 12, 0, "taps"),
    "pack_kernel": ("bf16 / fp16 weight shadow refresh", "hbm", core_floats * 4 + core_floats * 2 * 3, 0, "fp32 arena; 16-bit planes + transposes"),
}
ks = d["kernels"]
# SQ counters (tools/pmc_sq_all.sh), if collected: MFMA-pipe busy fraction, calibrated on the bench's own back-to-back MFMA probe kernel in the same pass
# (SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES of a kernel over the same ratio of mfma_peak_kernel, whose pipes are busy every cycle)
sq = {}
sqp = os.path.join(ROOT, "profiles", f"{rnd}_pmc_sq_all{tag}.json")
if os.path.exists(sqp):
    sk = json.load(open(sqp))["kernels"]
    ref = sk.get("mfma_peak_kernel")
    if ref and ref.get("SQ_BUSY_CYCLES"):
        r0 = ref["SQ_VALU_MFMA_BUSY_CYCLES"] / ref["SQ_BUSY_CYCLES"]
        for k_, e_ in sk.items():
            if e_.get("SQ_BUSY_CYCLES"):
                sq[k_] = {"mfma_busy": e_["SQ_VALU_MFMA_BUSY_CYCLES"] / e_["SQ_BUSY_CYCLES"] / r0, "wait": e_["wait_any_frac"], "stall": e_["wait_inst_frac"], "valu": e_["active_valu_frac"]}
steps = d.get("bench_alone", {}).get("steps", 3) + d.get("bench_alone", {}).get("warmup", 2)
fw = next((e for k_, e in ks.items() if k_.startswith("attn_fwd_kernel") and e.get("calls_alone")), None)
if fw:  # bench.py also runs un-timed steps (the roofline_hbm window): count the steps of the pass by the forward attention's 4 launches per step
    steps = fw["calls_alone"] / 4.0
step_ms_alone = sum(e.get("total_ms_alone", 0.0) for k, e in ks.items() if "mfma_peak" not in k) / steps
rows = []
for k, e in ks.items():
    if "mfma_peak" in k or e.get("total_ms_alone", 0) / steps < 0.003 * step_ms_alone or "fetch_kib" not in e:
        continue
    traffic = (2 * e["fetch_kib"] + e.get("write_kib", 0.0)) * 1024
    us, live = e["avg_us_alone"], e.get("avg_us_live", 0.0)
    m = MODEL.get(k)
    rows.append((e["total_ms_alone"] / steps, k, e["calls_alone"] / steps, us, live, 2 * e["fetch_kib"] * 1024, e.get("write_kib", 0.0) * 1024, traffic, m))
rows.sort(reverse=True)
out = [f"# Counter-backed roofline of every kernel of the C2 training step ({rnd})", "",
       f"Source: `profiles/{rnd}_pmc_all{tag}.json` (`tools/pmc_all.sh`: separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, KiB per launch; durations from",
       "`--kernel-trace --stats` passes, *alone* = `V1T_DW_SIDE=0` (every kernel has the chip), *live* = as the step runs (weight-gradient GEMMs beside the main stream).",
       "traffic = FETCH_SIZE x 2 + WRITE_SIZE (the guide's gfx950 correction; the counters sit on the L2's fabric side, Infinity-Cache hits included). The x 2 is",
       f"calibrated per kernel here: `profiles/{rnd}_pmc_readsize.json` (`tools/pmc_readsize.sh`: TCC_EA0_RDREQ by request size) shows EVERY read request of every kernel of",
       "the step is a 128-B request (32-B: 0, 64-B: < 0.3 %), which FETCH_SIZE tallies at 64 B - so the doubling holds for the strided 16-B row reads of `ln_gemm` / `mlp_fwd` too,",
       "whose 2 x 113 MiB for a 113-MiB residual stream is a real second fabric fetch of half of every line (8 waves x 20 KB of rows in flight per CU thrash the 16-KB L1 and the XCD's L2).",
       "TB/s = traffic / alone duration; ceilings on this chip (profiles/r04_hbm_stream.txt): read 6.4, write 4.7, copy 5.2 TB/s, nominal 8.",
       f"Kernels below 0.3 % of the step omitted. Kernel time of one step (alone, sum): {step_ms_alone:.2f} ms.", "",
       "MFMA busy = fraction of cycles the matrix pipes are busy (`profiles/" + rnd + "_pmc_sq_all.json`, calibrated on the back-to-back MFMA probe of the same pass); wait / stall = share of wave",
       "cycles parked at s_waitcnt / barriers and stalled at issue (SQ_WAIT_ANY, SQ_WAIT_INST_ANY).", "",
       "| kernel | what | per step | alone us | live us | read GB | written GB | TB/s | of 8 | of ceiling | alg. GB | traffic / alg. | TFLOP/s | MFMA busy | wait / stall | bound |",
       "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
js = {}
for ms, k, calls, us, live, rd, wr, traffic, m in rows:
    tbs = traffic / (us * 1e-6) / 1e12
    ceil = (6.4 * rd + 4.7 * wr) / max(rd + wr, 1)  # traffic-weighted streaming ceiling
    what, bound, alg, fl, tensors = m if m else ("", "?", 0, 0, "")
    ratio = traffic / alg if alg else float("nan")
    tf = fl / (us * 1e-6) / 1e12 if fl else 0.0
    out.append(f"| `{k}` | {what} | {calls:.1f} x = {ms:.2f} ms | {us:.0f} | {live:.0f} | {rd / 1e9:.2f} | {wr / 1e9:.2f} | {tbs:.2f} | {tbs / 8:.2f} | {tbs / ceil:.2f} | "
               f"{alg / 1e9:.2f} | {ratio:.2f} | {tf:.0f} | {sq.get(k, {}).get('mfma_busy', float('nan')):.2f} | {sq.get(k, {}).get('wait', float('nan')):.2f} / {sq.get(k, {}).get('stall', float('nan')):.2f} | {bound} |")
    js[k] = {"per_step_ms": ms, "alone_us": us, "live_us": live, "read_bytes": rd, "written_bytes": wr, "tbps": tbs, "frac_of_8": tbs / 8, "frac_of_ceiling": tbs / ceil,
             "algorithmic_bytes": alg, "traffic_over_algorithmic": ratio, "tflops": tf, "bound": bound, "tensors": tensors, **{"sq_" + a_: b_ for a_, b_ in sq.get(k, {}).items()}}
out += ["", "Tensors counted as algorithmic bytes:", ""] + [f"* `{k}`: {v['tensors']}" for k, v in js.items() if v["tensors"]]
open(os.path.join(ROOT, "profiles", f"{rnd}_roofline_table{tag}.md"), "w").write("\n".join(out) + "\n")
json.dump(js, open(os.path.join(ROOT, "profiles", f"{rnd}_roofline_table{tag}.json"), "w"), indent=1)
print("\n".join(out[:9] + [r[:230] for r in out[9:9 + len(rows)]]))
