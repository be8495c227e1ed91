"""Builds libdeftet_hip.so in-tree with hipcc for gfx950 (no JIT cache, no torch headers).

    python -m deftet_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdeftet_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# -ffp-contract=off: the integer-valued outputs (tet index, argmin face, NN index) are
# decided by fp32 sign tests / comparisons that must follow the reference's operation
# order without fused multiply-adds ("numerics").
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _deps():
    out = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    out.append(os.path.join(os.path.dirname(HERE), "include", "deftet_hip.h"))
    return out


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in _deps())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    if not os.path.exists(HIPCC):
        raise RuntimeError("hipcc not found at %s — cannot build libdeftet_hip.so" % HIPCC)
    srcs = sources()
    cmd = [HIPCC] + FLAGS + ["-x", "hip"] + srcs + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
