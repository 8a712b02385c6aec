"""Builds tests/fake_rccl.cpp (TEST INFRASTRUCTURE: a same-device transport stand-in for the ten RCCL
entry points the product binds with dlopen) and hands out the environment that makes the product load it
(EPA_RCCL_LIB).  With it the product's gather protocol -- epa_ng_amd/csrc/comm.hip, host/place_ranks.cpp --
runs with world = 2, 3, 8 processes on ONE GPU (RCCL itself refuses two ranks on one device)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "fake_rccl.cpp")
OUT = os.path.join(HERE, "_build", "libfake_rccl.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def build(force=False):
    abi = os.path.join(os.path.dirname(HERE), "epa_ng_amd", "csrc", "rccl_abi.hpp")
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(SRC), os.path.getmtime(abi)):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROCM, "include"),
           "-Wall", "-Wextra", SRC, "-o", OUT + ".tmp", "-L", os.path.join(ROCM, "lib"), "-lamdhip64",
           "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-lpthread"]
    subprocess.check_call(cmd)
    os.replace(OUT + ".tmp", OUT)
    return OUT


def env(base=None, timeout_s=120):
    """environment of a rank process that should run the product's gather over the stand-in"""
    e = dict(os.environ if base is None else base)
    e["EPA_RCCL_LIB"] = build()
    e["EPA_FAKE_RCCL_TIMEOUT_S"] = str(timeout_s)
    e["EPA_COMM_TIMEOUT_S"] = str(timeout_s + 30)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return e


SYMBOLS = ["ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclCommAbort", "ncclSend", "ncclRecv",
           "ncclGroupStart", "ncclGroupEnd", "ncclAllReduce", "ncclGetErrorString"]
