"""Build libv1t_amd.so (gfx950) in-tree with hipcc. `python -m v1t_amd.build [--force]`.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the
repo snapshot. No torch / pybind linkage: the library exposes the plain C-ABI of include/v1t_amd.h.
"""
from __future__ import annotations

import hashlib
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libv1t_amd.so")
SOURCES = ["api.hip", "gemm.hip", "attention.hip", "elementwise.hip", "readout.hip", "gridprep.hip", "metrics.hip", "data.hip", "probe.hip", "tails.hip"]
HEADERS = ["common.h", "gemm.h", "attention.h", "elementwise.h", "readout.h", "gridprep.h", os.path.join("..", "..", "include", "v1t_amd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
EXTRA = os.environ.get("V1T_HIPCC_EXTRA", "").split()  # dev: extra compiler flags (ablations)
# attention: no SLP vectorisation - packed f32 VALU (v_pk_mul_f32 / v_pk_fma_f32) beside MFMAs issue slower than the two scalar
# instructions they replace (MI355X_MICROARCH.md; A/B in one process: forward -3.8 %, backward -1 %)
PER_FILE = {"attention.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _sha(paths, extra=()) -> str:
    h = hashlib.sha256()
    for p_ in paths:
        with open(p_, "rb") as fh:
            h.update(hashlib.sha256(fh.read()).digest())
    for e in extra:
        h.update(str(e).encode() + b"\0")
    return h.hexdigest()


def all_sources() -> list:
    """Every file the library is compiled from: ALL .hip / .h under csrc/ (not a hand-kept list) and the public header."""
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    return [os.path.join(CSRC, f) for f in files] + [os.path.join(HERE, "..", "include", "v1t_amd.h")]


EXP_LIB = os.path.join(LIBDIR, "libv1t_amd_exp.so")  # the experiment build: the same sources with -DV1T_EXPERIMENTS (dev switches + experiment kernels)
EXP_FLAGS = ["-DV1T_EXPERIMENTS"]


def _extra_for(lib_path: str) -> list:
    return EXP_FLAGS if os.path.basename(lib_path) == os.path.basename(EXP_LIB) else EXTRA


def sources_sha16(extra=None) -> str:
    """Content hash over every source + header + the compiler flags: what `buildinfo` records and `is_current` compares. Staleness is by
    CONTENT, not by mtime: the git-ignored .so travels with the repo snapshot, and a restored tree or a skewed clock must not run the
    tests against a library built from other sources."""
    return _sha(all_sources(), [FLAGS, sorted(PER_FILE.items()), EXTRA if extra is None else extra])[:16]


def attention_sha16() -> str:
    return _sha_plain([os.path.join(CSRC, f) for f in ("attention.hip", "attention.h", "common.h")])[:16]


def _sha_plain(paths) -> str:
    h = hashlib.sha256()  # the form bench.py / tools/pmc_bench.sh have used since round 4 for the attention sources (file contents back to back)
    for p_ in paths:
        with open(p_, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def buildinfo(lib_path: str = LIB) -> dict:
    try:
        with open(lib_path + ".buildinfo.json") as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return {}


def is_current(lib_path: str = LIB) -> bool:
    """The library exists and was built from exactly the sources (and flags) in the tree."""
    return os.path.exists(lib_path) and buildinfo(lib_path).get("sources_sha16") == sources_sha16(_extra_for(lib_path))


def build_experiments(force: bool = False, verbose: bool = True) -> str:
    """libv1t_amd_exp.so: the product's sources compiled with -DV1T_EXPERIMENTS - the development switches (csrc/common.h `dev_env`,
    tools/DEV_SWITCHES.md) and the experiment kernels of rounds 1-5 exist only here. The equality tests (fused == unfused forms) load it
    in a child process with V1T_LIB; nothing else does."""
    return build(force=force, verbose=verbose, alt=os.path.basename(EXP_LIB), extra=EXP_FLAGS)


def build(force: bool = False, verbose: bool = True, alt: str = None, extra: list = None) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    EXTRA = globals()["EXTRA"] if extra is None else list(extra)
    hdrs = [f for f in all_sources() if f.endswith(".h")]
    listed = {os.path.join(CSRC, s_) for s_ in SOURCES}
    stray = [f for f in all_sources() if f.endswith(".hip") and f not in listed]
    if stray:
        raise RuntimeError(f"csrc/ holds .hip files that are not in build.SOURCES: {stray}")
    hipcc = _hipcc()
    objs, jobs = [], []
    # dev: V1T_BUILD_LIB=libv1t_amd_x.so builds a second library next to the product one (its own objects; V1T_HIPCC_EXTRA flags),
    # loaded with V1T_LIB=libv1t_amd_x.so - A/B experiments and the in-kernel probe builds
    alt = os.environ.get("V1T_BUILD_LIB", "") if alt is None else alt
    ablation = ("V1T_DEV_ABLATION", "V1T_F3_", "V1T_B2_")
    if not alt and any(any(a in f for a in ablation) for f in EXTRA):
        # timing-only ablations compile pieces of kernels out and return garbage: never into the product library
        raise RuntimeError("V1T_DEV_ABLATION* flags build experiment libraries only: set V1T_BUILD_LIB=libv1t_amd_<name>.so")
    lib_path = os.path.join(LIBDIR, alt) if alt else LIB
    objdir = os.path.join(LIBDIR, alt + ".objs") if alt else LIBDIR
    os.makedirs(objdir, exist_ok=True)
    stamps = {}
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        want = _sha([sp] + hdrs, [FLAGS, PER_FILE.get(src, []), EXTRA])  # an object depends on its source, every header and its flags
        try:
            have = open(obj + ".sha").read().strip()
        except OSError:
            have = ""
        if force or not os.path.exists(obj) or have != want:
            jobs.append([hipcc, *FLAGS, *PER_FILE.get(src, []), *EXTRA, "-c", sp, "-o", obj])
            stamps[obj] = want

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(4, (os.cpu_count() or 8) // 2))) as ex:
            list(ex.map(run, jobs))
        for obj, want in stamps.items():
            with open(obj + ".sha", "w") as fh:
                fh.write(want)
    if jobs or force or not is_current(lib_path):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path, *objs])
        # what this library was built from: every source (is_current, the smoke's report) and the attention sources the tracked PMC summary
        # was measured on (bench.py)
        with open(lib_path + ".buildinfo.json", "w") as fh:
            json.dump({"sources_sha16": sources_sha16(EXTRA), "attention_sources_sha16": attention_sha16(), "extra_flags": EXTRA,
                       "n_sources": len(all_sources())}, fh)
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    if "--experiments" in sys.argv:
        print(build_experiments(force="--force" in sys.argv))
