"""Builds multiply_amd/libmultiply_hip.so from csrc/*.hip with hipcc for gfx950 (in-tree, so it travels with the repo)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# MP_BUILD_TAG=<name> (+ MP_EXTRA_FLAGS=-D...): an ablation side library multiply_amd/ab_libs/libmultiply_hip_<name>.so
# with its own object directory; load it with MP_LIB_PATH (tools/ab_run.sh).  Unset: the product library.
TAG = os.environ.get("MP_BUILD_TAG", "")
OUT = os.path.join(HERE, "ab_libs", f"libmultiply_hip_{TAG}.so") if TAG else os.path.join(HERE, "libmultiply_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"] + os.environ.get("MP_EXTRA_FLAGS", "").split()


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "hipcc")
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "multiply_hip.h"))
    bdir = os.path.join(CSRC, "_build" + ("_" + TAG if TAG else ""))
    os.makedirs(bdir, exist_ok=True)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    jobs = []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(bdir, s[:-4] + ".o")
        if force or _newer([src] + hdrs, obj):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(bdir, s[:-4] + ".o") for s in srcs]
    if force or jobs or _newer(objs, OUT):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
