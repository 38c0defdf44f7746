"""Builds tests/rccl_shim.cpp (TEST INFRASTRUCTURE: RCCL's entry points between processes that share one GPU) into
tests/_build/librccl_shim.so and returns its path.  Loaded by the product through R3N_RCCL_LIB (rend3_amd/csrc/comm.h)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "rccl_shim.cpp")
OUT = os.path.join(HERE, "_build", "librccl_shim.so")
SYMBOLS = ["ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclGetErrorString", "ncclGroupStart", "ncclGroupEnd",
           "ncclBroadcast", "ncclAllGather", "ncclAllReduce", "ncclReduceScatter"]


def build():
    if os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    tmp = OUT + f".{os.getpid()}.tmp"  # two ranks may build at once: each writes its own file, the rename is atomic
    res = subprocess.run([hipcc, "-O2", "-std=c++17", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", tmp, SRC, "-lrt"], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("rccl_shim build failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, OUT)
    return OUT
