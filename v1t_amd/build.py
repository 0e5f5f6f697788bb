"""Build libv1t_amd.so (gfx950) in-tree with hipcc. `python -m v1t_amd.build [--force]`.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the
repo snapshot. No torch / pybind linkage: the library exposes the plain C-ABI of include/v1t_amd.h.
"""
from __future__ import annotations

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


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    hipcc = _hipcc()
    objs, jobs = [], []
    # dev: V1T_BUILD_LIB=libv1t_amd_x.so builds a second library next to the product one (its own objects; V1T_HIPCC_EXTRA flags),
    # loaded with V1T_LIB=libv1t_amd_x.so - A/B experiments and the in-kernel probe builds
    alt = os.environ.get("V1T_BUILD_LIB", "")
    ablation = ("V1T_DEV_ABLATION", "V1T_F3_", "V1T_B2_")
    if not alt and any(any(a in f for a in ablation) for f in EXTRA):
        # timing-only ablations compile pieces of kernels out and return garbage: never into the product library
        raise RuntimeError("V1T_DEV_ABLATION* flags build experiment libraries only: set V1T_BUILD_LIB=libv1t_amd_<name>.so")
    lib_path = os.path.join(LIBDIR, alt) if alt else LIB
    objdir = os.path.join(LIBDIR, alt + ".objs") if alt else LIBDIR
    os.makedirs(objdir, exist_ok=True)
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [sp] + hdrs):
            jobs.append([hipcc, *FLAGS, *PER_FILE.get(src, []), *EXTRA, "-c", sp, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(lib_path, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path, *objs])
        # what this library was built from: bench.py compares it with the sources the tracked PMC summary was measured on
        import hashlib
        import json

        h = hashlib.sha256()
        for f in ("attention.hip", "attention.h", "common.h"):
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(fh.read())
        with open(lib_path + ".buildinfo.json", "w") as fh:
            json.dump({"attention_sources_sha16": h.hexdigest()[:16], "extra_flags": EXTRA}, fh)
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
