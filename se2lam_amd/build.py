"""Build recipe of libse2gpu.so (HIP kernels + C ABI) for gfx950, in-tree.

    python -m se2lam_amd.build [--force]

Each csrc/*.hip is compiled to an object with hipcc (--offload-arch=gfx950) and linked into
se2lam_amd/lib/libse2gpu.so.  The ORB / matcher translation units are compiled with
-ffp-contract=off: their float index arithmetic (x*b + y*a, fastAtan2) must round exactly like
the reference's non-FMA x86 build so descriptors and matches are bit-exact (SURVEY.md §7 hard
part 3).  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libse2gpu.so")
OBJDIR = os.path.join(LIBDIR, "obj")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-Wno-unused-result", "-I", INCLUDE]
PER_FILE = {
    "orb.hip": ["-ffp-contract=off"],
    "match.hip": ["-ffp-contract=off"],
    "triangulate.hip": ["-ffp-contract=off"],
    "ransac.hip": ["-ffp-contract=off"],
    "sparsify.hip": ["-ffp-contract=off"],   # forward differences with delta 1e-6 decide the result: reproduce them to the bit
}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    headers.append(os.path.abspath(__file__))
    objs = []
    for src in sources():
        obj = os.path.join(OBJDIR, src[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [os.path.join(CSRC, src)] + headers):
            cmd = [_hipcc()] + COMMON + PER_FILE.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
