"""Build libspgnn_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(os.path.dirname(HERE), "libspgnn_hip.so")
SOURCES = [os.path.join(HERE, "spgnn_kernels.hip"), os.path.join(HERE, "spgnn_gemm.hip")]


def build(force: bool = False, verbose: bool = True) -> str:
    deps = SOURCES + [os.path.join(ROOT, "include", "spgnn_hip.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC",
           "-I", os.path.join(ROOT, "include"), "-o", OUT] + SOURCES
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
