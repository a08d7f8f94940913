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
SOURCES = ["nsol_blur3_f32.hip", "nsol_blur3_lz_f32.hip", "nsol_blur3_f64.hip", "nsol_blur3_lz_f64.hip", "nsol_conv.hip", "nsol_ops.hip", "nsol_pd.hip", "nsol_pd2.hip", "nsol_pdk.hip", "nsol_pdp.hip",
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


def build_library(force=False, verbose=False, jobs=None):
    """Compile the HIP sources into csrc/libnsol_hip.so; returns its path.
    The translation units are compiled side by side (objects in a temporary
    directory outside the tree) and linked once."""
    if not force and not _stale():
        return LIB
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    compile_flags = [f for f in FLAGS if f != "-shared"]
    jobs = jobs or min(len(SOURCES), max(1, (os.cpu_count() or 2) - 2))
    tmp = tempfile.mkdtemp(prefix="nsol_build_")
    try:
        def one(src):
            obj = os.path.join(tmp, src.replace(".hip", ".o"))
            cmd = [hipcc] + compile_flags + ["-I", INCLUDE, "-c",
                                             os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
            return obj
        with ThreadPoolExecutor(max_workers=jobs) as pool:
            objs = list(pool.map(one, SOURCES))
        cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + \
            ["-o", LIB + ".tmp"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        os.replace(LIB + ".tmp", LIB)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
