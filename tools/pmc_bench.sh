# usage (GPU box): bash tools/pmc_bench.sh  -> gpurun_out/${RND:-r06}_pmc_attention.json + .txt: per-launch FETCH_SIZE / WRITE_SIZE (KiB) of the
# attention kernels inside bench.py (112-image training launches, dropout on) and inside the C5 eval pass (256-image launches),
# separate --pmc passes as MI355X_MICROARCH.md prescribes; FETCH_SIZE needs x2 on gfx950 (applied by bench.py, not here).
# Copy the two files to profiles/ (tracked) - bench.py reads profiles/${RND:-r06}_pmc_attention.json at run time.
export RND=${RND:-r06}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmcb_$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-pmc > /tmp/pmcb_$C.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmce_$C -- python3 $GRAFT_REPO_ROOT/bench.py --config c5 --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > /tmp/pmce_$C.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, os, collections
import hashlib
_h = hashlib.sha256()
for _f in ("attention.hip", "attention.h", "common.h"):  # the same hash bench.py computes (sources_hash): a later kernel change shows as traffic_stale
    _h.update(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "v1t_amd", "csrc", _f), "rb").read())
out = {"shape": {"H": 4, "T": 1654, "DP": 160}, "sources_sha16": _h.hexdigest()[:16], "kernels": {}, "source": "tools/pmc_bench.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB per launch, mean over launches"}
names = {"attn_bwd_dkv2_kernel": "attn_bwd_dkv2", "attn_bwd_dq2_kernel": "attn_bwd_dq2", "attn_fwd_kernel": "attn_fwd", "attn_fwd2_kernel": "attn_fwd", "attn_delta2_kernel": "attn_delta2"}
lines = []
for tag, images, suffix in (("pmcb", 112, ""), ("pmce", 256, "_eval")):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"/tmp/{tag}_{c}/*/*counter_collection.csv")
        if not f:
            print("no counter file for", tag, c); continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if r.get("Counter_Name") == c and "attn" in r["Kernel_Name"]:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            short = next((n for s, n in names.items() if s in k), None)
            if short is None or (suffix and short != "attn_fwd"):
                continue
            e = out["kernels"].setdefault(short + suffix, {"images": images})
            e["fetch_kib" if c == "FETCH_SIZE" else "write_kib"] = sum(v) / len(v)
            lines.append(f"{c:10s} {k[:80]:80s} images/launch={images:4d} n={len(v):3d} mean={sum(v)/len(v):14.1f} min={min(v):14.1f} max={max(v):14.1f}")
os.makedirs(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", os.environ["RND"] + "_pmc_attention.json"), "w"), indent=1)
open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", os.environ["RND"] + "_pmc_attention_fetch_write.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines)); print(json.dumps(out))
PY
