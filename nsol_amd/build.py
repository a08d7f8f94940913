"""Builds libnsol_hip.so (gfx950) in-tree with hipcc.

`python -m nsol_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles
without a GPU, so this runs in the CPU-only build container too.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(HERE, "..", "include")
LIB = os.path.join(CSRC, "libnsol_hip.so")
SOURCES = ["nsol_ops.hip", "nsol_conv.hip", "nsol_pd.hip", "nsol_pd2.hip", "nsol_pdk.hip",
           "nsol_lsmr.hip", "nsol_lbfgsb.hip", "nsol_sort.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
         "-fPIC", "-shared"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)
            if f.endswith((".hip", ".hpp"))]
    deps.append(os.path.join(INCLUDE, "nsol_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile the HIP sources into csrc/libnsol_hip.so; returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + ["-I", INCLUDE] + \
        [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
