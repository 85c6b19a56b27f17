"""Build libspgnn_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU).

Each .hip source is compiled to its own object (in parallel: the three files take 30-80 s each) and only the
sources that changed since the last build are recompiled; then one link step."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(os.path.dirname(HERE), "libspgnn_hip.so")
OBJ_DIR = os.path.join(ROOT, "build", "obj")
SOURCES = [os.path.join(HERE, n) for n in ("spgnn_kernels.hip", "spgnn_gemm.hip", "spgnn_bf16.hip")]
HEADERS = [os.path.join(ROOT, "include", "spgnn_hip.h"), os.path.join(HERE, "spgnn_internal.h")]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", HERE]
# Per-source flags.  The row kernels are built WITHOUT the SLP vectorizer: the packed fp32 ops it forms (v_pk_fma_f32 fed by
# v_pk_mov_b32 op_sel shuffles) gave transiently wrong per-edge dots in gat_bwd_dst when a second process shared the GPU
# (spgnn_kernels.hip, SPGNN_DIST_DST; tools/dbg_dst_repro.py); scalar fp32 code is as fast there.  The GEMM files keep it:
# their packed conversions are what the split costs least with, and their cross-lane reads are guarded by hand.
EXTRA_FLAGS = {"spgnn_kernels.hip": ["-fno-slp-vectorize", "-DSPGNN_NO_SLP_VECTORIZE"]}


def _obj(src: str) -> str:
    return os.path.join(OBJ_DIR, os.path.basename(src) + ".o")


def _stale(target: str, deps) -> bool:
    return not os.path.exists(target) or any(os.path.getmtime(target) < os.path.getmtime(d) for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    todo = [s for s in SOURCES if force or _stale(_obj(s), [s, os.path.abspath(__file__)] + HEADERS)]

    def compile_one(src):
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if todo:
        with ThreadPoolExecutor(max_workers=len(todo)) as ex:
            list(ex.map(compile_one, todo))
    objs = [_obj(s) for s in SOURCES]
    if todo or _stale(OUT, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
