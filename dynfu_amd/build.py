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
SOURCES = ["tsdf.hip", "warp.hip", "solve.hip", "solve6.hip", "mc.hip", "img.hip", "icp.hip", "points.hip", "capi.cpp"]
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


DEV_LIB = os.path.join(HERE, "libdynfu_amd_dev.so")


def build(force=False, verbose=False, dev=False):
    """dev=False: the product library (no A/B switch, no non-default kernel variant in it).  dev=True: the same sources
    with -DDFA_DEV_AB -> dynfu_amd/libdynfu_amd_dev.so (csrc/dev_switch.hpp): what the tests that compare kernel variants
    and the A/B scripts under tools/ load (dynfu_amd._lib.use_library)."""
    obj_dir = os.path.join(OBJ, "dev") if dev else OBJ
    lib = DEV_LIB if dev else LIB
    flags = FLAGS + (["-DDFA_DEV_AB"] if dev else [])
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(HERE, "..", "include", "dynfu_amd.h"))
    headers.append(os.path.abspath(__file__))
    cc = hipcc()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, src.rsplit(".", 1)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [cc] + flags + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", s, "-o", o]
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
    if force or jobs or _stale(lib, objs):
        run([cc, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", lib])
    return lib


HOST = os.path.join(HERE, "host")
HOST_LIB = os.path.join(HERE, "libdynfu_amd_host.so")
HOST_SOURCES = ["device.cpp", "frame.cpp", "tsdf_volume.cpp", "warp_field.cpp", "opt_solver.cpp", "northstar_solver.cpp", "dyn_fusion.cpp",
                "marching_cubes.cpp", "imgproc.cpp", "projective_icp.cpp", "io.cpp", "kinfu.cpp"]


def build_host(force=False, verbose=False):
    """C++ host adaptors (kfusion::cuda::TsdfVolume, Warpfield, CombinedSolver, ...) above the C ABI
    -> dynfu_amd/libdynfu_amd_host.so.  Plain C++ (no device code), linked against
    libdynfu_amd.so and the HIP runtime."""
    lib = build(force=force, verbose=verbose)
    inc = os.path.join(HOST, "include")
    srcs = [os.path.join(HOST, "src", f) for f in HOST_SOURCES]
    deps = srcs + [lib, os.path.abspath(__file__)]
    for root, _, files in os.walk(inc):
        deps += [os.path.join(root, f) for f in files]
    if force or _stale(HOST_LIB, deps):
        cmd = [hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I" + inc] + srcs + \
              ["-L" + HERE, "-ldynfu_amd", "-lz", "-Wl,-rpath,$ORIGIN", "-o", HOST_LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("host library build failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
    return HOST_LIB


SEQ_BENCH = os.path.join(HERE, "host", "build", "sequence_bench")


def build_host_tools(force=False, verbose=False):
    """dynfu_amd/host/tools/sequence_bench.cpp -> dynfu_amd/host/build/sequence_bench: DynFusion::operator() frame by
    frame over raw depth frames (bench.py's end_to_end figure)."""
    host = build_host(force=force, verbose=verbose)
    src = os.path.join(HOST, "tools", "sequence_bench.cpp")
    os.makedirs(os.path.dirname(SEQ_BENCH), exist_ok=True)
    if force or _stale(SEQ_BENCH, [src, host]):
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(HOST, "include"), "-I/opt/rocm/include",
               "-D__HIP_PLATFORM_AMD__", src, "-o", SEQ_BENCH, "-L" + HERE, "-ldynfu_amd_host", "-ldynfu_amd",
               "-Wl,-rpath," + HERE, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("sequence_bench build failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
    return SEQ_BENCH


def build_cpp_tests(force=False, verbose=False):
    """tests/cpp/*.cpp -> tests/cpp/build/<name> (git-ignored; travels to the GPU box)."""
    host = build_host(force=force, verbose=verbose)
    root = os.path.dirname(HERE)
    tdir = os.path.join(root, "tests", "cpp")
    out = os.path.join(tdir, "build")
    os.makedirs(out, exist_ok=True)
    inc = os.path.join(HOST, "include")
    oracle_dir = os.path.join(root, "oracle")
    built = {}
    for name, needs in (("test_host_dq", []), ("test_host_solver", ["host"]), ("test_host_tsdf", ["host", "oracle"]),
                        ("test_host_dynfusion", ["host", "oracle"]), ("test_host_icp", ["host"]),
                        ("test_tsdf_classify", ["oracle", "ieee"]), ("test_host_io", ["host"]), ("test_host_kinfu", ["host"])):
        src = os.path.join(tdir, name + ".cpp")
        exe = os.path.join(out, name)
        deps = [src, os.path.join(tdir, "minitest.hpp"), host, os.path.join(CSRC, "tsdf_classify.hpp")]
        if force or _stale(exe, deps):
            cmd = ["g++", "-O1", "-std=c++17", "-Wall", "-I" + inc, src, "-o", exe]
            if "ieee" in needs:  # CPU model of a kernel: same arithmetic contract as the oracle, hardware fma
                cmd[1:2] = ["-O2", "-march=x86-64-v3", "-ffp-contract=off"]
            if name == "test_host_io":  # its `sequence` mode drives DynFusion (GPU); the test cases themselves are CPU-only
                cmd.insert(1, "-DDFA_WITH_DYNFUSION")
            if "host" in needs:
                cmd += ["-L" + HERE, "-ldynfu_amd_host", "-ldynfu_amd", "-Wl,-rpath," + HERE,
                        "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"]
            if "oracle" in needs:
                cmd += ["-L" + oracle_dir, "-loracle", "-Wl,-rpath," + oracle_dir]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("C++ test build failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
        built[name] = exe
    return built


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--dev" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, dev=True))
    if "--host" in sys.argv:
        print(build_host(force="--force" in sys.argv, verbose=True))
        print(build_cpp_tests(force="--force" in sys.argv, verbose=True))
