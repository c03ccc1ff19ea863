"""Builds multiply_amd/libmultiply_hip.so from csrc/*.hip with hipcc for gfx950 (in-tree, so it travels with the repo).

Freshness is decided by CONTENT, not by modification time: every object carries a stamp `<obj>.sha` = sha256 over its source,
every header and the compiler flags, and the library a stamp `<lib>.sha16` = sha256 over all of those (`source_hash()`).  An
object / the library is reused only when its stamp equals the hash of the tree being built -- a checkout that leaves old objects
with newer time stamps than edited sources cannot link stale code.  `build()` returns the library path; `last_build()` says
whether the last call compiled anything and which source hash the library carries (printed by __graft_entry__.build() and
recorded by bench.py as `lib_source_sha16`).
"""
import hashlib
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
_LAST = {"mode": None, "sha16": None, "compiled": []}


def _sources():
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp"))
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "multiply_hip.h"))
    return srcs, hdrs


def _digest(paths, extra=()):
    h = hashlib.sha256()
    for e in extra:
        h.update(e.encode())
        h.update(b"\0")
    for p in paths:
        h.update(os.path.basename(p).encode())
        h.update(b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def source_hash():
    """sha256 (16 hex digits) over every csrc/*.hip, every header and the compiler flags: what the library is built from"""
    srcs, hdrs = _sources()
    return _digest([os.path.join(CSRC, s) for s in srcs] + hdrs, FLAGS)[:16]


def _stamp(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def library_hash():
    """the source hash recorded next to the built library (None: no library / no stamp)"""
    return _stamp(OUT + ".sha16") if os.path.exists(OUT) else None


def last_build():
    return dict(_LAST)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "hipcc")
    srcs, hdrs = _sources()
    bdir = os.path.join(CSRC, "_build" + ("_" + TAG if TAG else ""))
    os.makedirs(bdir, exist_ok=True)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    want_lib = source_hash()
    jobs = []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(bdir, s[:-4] + ".o")
        want = _digest([src] + hdrs, FLAGS)
        if force or not os.path.exists(obj) or _stamp(obj + ".sha") != want:
            jobs.append(([hipcc] + FLAGS + ["-c", src, "-o", obj], obj, want))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)

    def compile_one(job):
        cmd, obj, want = job
        if os.path.exists(obj + ".sha"):
            os.remove(obj + ".sha")
        run(cmd)
        with open(obj + ".sha", "w") as f:
            f.write(want + "\n")

    with ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 1))) as ex:
        list(ex.map(compile_one, jobs))
    objs = [os.path.join(bdir, s[:-4] + ".o") for s in srcs]
    relink = force or bool(jobs) or not os.path.exists(OUT) or _stamp(OUT + ".sha16") != want_lib
    if relink:
        if os.path.exists(OUT + ".sha16"):
            os.remove(OUT + ".sha16")
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
        with open(OUT + ".sha16", "w") as f:
            f.write(want_lib + "\n")
    _LAST.update(mode="built" if relink else "reused", sha16=want_lib, compiled=[os.path.basename(j[1]) for j in jobs])
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT, last_build())
