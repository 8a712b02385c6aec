"""Builds the in-tree native libraries for gfx950 (no JIT cache: the .so travels with the repo).

  libepa_dev.so   HIP kernels + C-ABI (include/epa_dev.h)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

DEV_SOURCES = ["epa_dev.hip", "preplace.hip", "thorough_dna.hip", "thorough_aa.hip", "thorough_aa_mfma.hip", "thorough_generic.hip", "comm.hip"]
# Per-file code-generation tuning, measured same-box (round 4, exp/ab.sh / ab_aa.sh, two interleaved rounds):
# the iterative ILP scheduler removes the dominant nucleotide kernel's scratch (44 -> 0 B per lane) and is worth
# 5.24 -> 5.17 ms per 262k-pair launch; the 20-state kernel 4.44 -> 4.37 ms per 25.7k pairs.  (iterative-minreg /
# -maxocc: slower; preplace.hip crashes this compiler's register allocator under the strategy: not used there.)
EXTRA_FLAGS = {"thorough_dna.hip": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
               "thorough_aa_mfma.hip": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]}
DEV_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
             "-I", CSRC, "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_dev(force=False, verbose=False):
    out = os.path.join(HERE, "libepa_dev.so")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, "epa_dev_internal.hpp"), os.path.join(CSRC, "wave_util.hpp"), os.path.join(CSRC, "rccl_abi.hpp"),
            os.path.join(ROOT, "include", "epa_dev.h")]
    objs, procs = [], []
    for src in DEV_SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(op)
        if force or _newer(op, [sp] + hdrs):
            cmd = [HIPCC] + DEV_FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((src, subprocess.Popen(cmd), [HIPCC] + DEV_FLAGS + ["-c", sp, "-o", op]))
    for src, p, plain in procs:
        if p.wait() != 0:
            # the per-file scheduling strategy is an optional tuning: fall back to the default flags
            if not EXTRA_FLAGS.get(src) or subprocess.call(plain) != 0:
                raise RuntimeError("hipcc failed on " + src)
    if force or procs or _newer(out, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out + ".tmp"] + objs
        subprocess.check_call(cmd)
        os.replace(out + ".tmp", out)   # atomic: a process that has the old file mapped keeps it
    return out


HOST_SOURCES = ["model.cpp", "aa_models.cpp", "parse_model.cpp", "tree.cpp", "place.cpp", "place_ranks.cpp", "io.cpp", "capi.cpp"]
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-fopenmp", "-Wall", "-Wextra", "-Wno-unused-parameter",
              "-I", os.path.join(ROOT, "include")]


def build_host(force=False, verbose=False):
    """libepa_host.so (C++ host pipeline above the C-ABI) and the epa-ng-amd CLI."""
    hdir = os.path.join(CSRC, "host")
    out = os.path.join(HERE, "libepa_host.so")
    exe = os.path.join(HERE, "epa-ng-amd")
    srcs = [os.path.join(hdir, s) for s in HOST_SOURCES]
    deps = srcs + [os.path.join(hdir, "epa_host.hpp"), os.path.join(ROOT, "include", "epa_dev.h")]
    link = ["-L", HERE, "-lepa_dev", "-Wl,-rpath,$ORIGIN"]
    if force or _newer(out, deps):
        cmd = ["g++"] + HOST_FLAGS + ["-shared", "-o", out + ".tmp"] + srcs + link
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        os.replace(out + ".tmp", out)
    main = os.path.join(hdir, "main.cpp")
    if force or _newer(exe, deps + [main, out]):
        cmd = ["g++"] + HOST_FLAGS + ["-o", exe + ".tmp", main, "-L", HERE, "-lepa_host", "-lepa_dev",
                                      "-Wl,-rpath,$ORIGIN"]
        subprocess.check_call(cmd)
        os.replace(exe + ".tmp", exe)
    return out


if __name__ == "__main__":
    print(build_dev(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
