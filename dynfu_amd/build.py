"""Builds dynfu_amd/libdynfu_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the snapshot.
Flags that are part of the numerics contract:
  -ffp-contract=off   fused multiply-adds only where the source says fmaf()
  (HIP's default correctly-rounded fp32 divide / sqrt is kept: no -ffast-math, no
   -fhip-fp32-correctly-rounded-divide-sqrt override)
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libdynfu_amd.so")
SOURCES = ["tsdf.hip", "warp.hip", "solve.hip", "capi.cpp"]
ARCH = "gfx950"
EXTRA = os.environ.get("DFA_EXTRA_CXXFLAGS", "").split()
FLAGS = EXTRA + ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(HERE, "..", "include", "dynfu_amd.h"))
    headers.append(os.path.abspath(__file__))
    cc = hipcc()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.rsplit(".", 1)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [cc] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", s, "-o", o]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([cc, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
