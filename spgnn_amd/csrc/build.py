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
SOURCES = [os.path.join(HERE, n) for n in ("spgnn_kernels.hip", "spgnn_lspe.hip", "spgnn_tile.hip", "spgnn_graph.hip", "spgnn_gemm.hip", "spgnn_bf16.hip", "spgnn_head.hip")]
HEADERS = [os.path.join(ROOT, "include", "spgnn_hip.h"), os.path.join(HERE, "spgnn_internal.h"), os.path.join(HERE, "spgnn_rows.h")]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", HERE]
# Per-source flags.  The row kernels are built WITHOUT the SLP vectorizer: the packed fp32 ops it forms (v_pk_fma_f32 fed by
# v_pk_mov_b32 op_sel shuffles) gave transiently wrong per-edge dots in gat_bwd_dst when a second process shared the GPU
# (spgnn_kernels.hip, SPGNN_DIST_DST; tools/dbg_dst_repro.py); scalar fp32 code is as fast there.  The GEMM files keep it:
# their packed conversions are what the split costs least with, and their cross-lane reads are guarded by hand.
# -fno-vectorize as well: the loop vectorizer paired the general-degree fallback loops into the same packed ops (645 of them in
# gat_fwd_vec / gat_agg_fwd); with both vectorizers off the object holds NO packed fp32 arithmetic, and _check_isa() below
# keeps it that way (the mechanism of the hazard is unconfirmed, so the fence is "no such instruction in this file").
_ROW_FLAGS = ["-fno-slp-vectorize", "-fno-vectorize", "-DSPGNN_NO_SLP_VECTORIZE"]
EXTRA_FLAGS = {"spgnn_kernels.hip": _ROW_FLAGS, "spgnn_lspe.hip": _ROW_FLAGS, "spgnn_tile.hip": _ROW_FLAGS}
NO_PACKED_FP32 = ("spgnn_kernels.hip", "spgnn_lspe.hip", "spgnn_tile.hip")     # objects that must not contain v_pk_{fma,mul,add}_f32 at all
LLVM_BIN = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def _obj(src: str) -> str:
    return os.path.join(OBJ_DIR, os.path.basename(src) + ".o")


def _stale(target: str, deps) -> bool:
    return not os.path.exists(target) or any(os.path.getmtime(target) < os.path.getmtime(d) for d in deps)


def device_isa(obj: str) -> str:
    """Disassembly of the gfx950 code object bundled in ``obj``."""
    import glob
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp(prefix="spgnn_isa_")
    try:
        local = os.path.join(tmp, "o.o")
        shutil.copy(obj, local)
        subprocess.check_call([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", local], cwd=tmp,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = glob.glob(local + ".*gfx950*")
        if not cos:
            raise RuntimeError(f"{obj}: no gfx950 code object found")
        return subprocess.check_output([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", cos[0]], text=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def packed_fp32_report(obj: str) -> dict:
    """{function: (packed fp32 ops, cross-lane reads)} for every function of ``obj`` that has packed fp32 arithmetic
    (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32); cross-lane reads = DPP modifiers, ds_bpermute / ds_swizzle, v_readlane,
    v_permlane."""
    import re
    fn, pk, xl = None, {}, {}
    for line in device_isa(obj).splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            fn = m.group(1)
            continue
        t = line.split()
        if fn is None or len(t) < 2:
            continue
        op = t[0]
        if re.match(r"v_pk_(fma|mul|add)_f32", op):
            pk[fn] = pk.get(fn, 0) + 1
        if "dpp" in line or op.startswith(("ds_bpermute", "ds_swizzle", "v_readlane", "v_permlane")):
            xl[fn] = xl.get(fn, 0) + 1
    return {f: (n, xl.get(f, 0)) for f, n in pk.items()}


_PACKED_F32 = ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_mov_b32")
HAZARD_WINDOW = 12      # instructions looked back from a cross-lane read for the writer of its source


def _vregs(tok: str):
    """Register numbers named by one operand token: 'v7' -> [7], 'v[4:5]' -> [4, 5]; anything else -> []."""
    import re
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return [int(m.group(1))]
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


def packed_crosslane_hazards(obj: str, window: int = HAZARD_WINDOW) -> list:
    """The instruction pattern behind DESIGN.md section 4.1's hazard, found in the ISA itself: a CROSS-LANE READ (a DPP
    operand, ds_bpermute / ds_swizzle data, v_readlane, v_permlane) whose source register was last written, within the
    previous ``window`` instructions, by a PACKED fp32 op (v_pk_fma / v_pk_mul / v_pk_add _f32, v_pk_mov_b32: two passes over
    the wave - the read got the wait states of a single-pass op and lanes 48-63 were read early).  A plain write in between
    (the v_mov_b32 of single_pass(), the register pinned by an empty asm) clears it.  -> [(function, reader, writer)]."""
    import re
    out = []
    fn, recent = None, []                       # recent: [(text, written vregs, is_packed)]
    for line in device_isa(obj).splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            fn, recent = m.group(1), []
            continue
        body = line.split("//")[0].strip()
        if fn is None or not body or body.endswith(":"):
            continue
        parts = body.split(None, 1)
        op = parts[0]
        ops_ = [t.strip() for t in parts[1].split(",")] if len(parts) > 1 else []
        srcs = []
        if "dpp" in op or " quad_perm:" in body or " row_" in body or " wave_" in body:
            if len(ops_) >= 2:
                srcs = _vregs(ops_[1].split()[0])                # the permuted operand is src0
        elif op.startswith(("ds_bpermute", "ds_permute")):
            if len(ops_) >= 3:
                srcs = _vregs(ops_[2].split()[0])                # vdst, vaddr, vdata
        elif op.startswith("ds_swizzle"):
            if len(ops_) >= 2:
                srcs = _vregs(ops_[1].split()[0])
        elif op.startswith(("v_readlane", "v_permlane", "v_readfirstlane")):
            for t in ops_[1:]:
                srcs += _vregs(t.split()[0])
        for r in srcs:
            for text, written, packed in reversed(recent[-window:]):
                if r in written:
                    if packed:
                        out.append((fn, body, text))
                    break
        written = _vregs(ops_[0].split()[0]) if ops_ and not op.startswith(("global_store", "buffer_store", "ds_write", "flat_store", "scratch_store", "s_")) else []
        recent.append((body, written, op.startswith(_PACKED_F32)))
        if len(recent) > 4 * window:
            del recent[:-window]
    return out


def _check_isa(src: str, verbose: bool, obj: str = None) -> None:
    """The fence behind DESIGN.md section 4.1's hazard: packed fp32 ops next to cross-lane reads gave transiently wrong
    values when a second process shared the GPU.  The row kernels' object must hold none at all (build error otherwise);
    for the GEMM objects, whose cross-lane sums are guarded by hand, the functions where both occur are listed."""
    rep = packed_fp32_report(obj or _obj(src))
    base = os.path.basename(src)
    if base in NO_PACKED_FP32:
        if rep:
            worst = sorted(rep.items(), key=lambda kv: -kv[1][0])[:5]
            raise RuntimeError(f"{base}: packed fp32 arithmetic in {len(rep)} function(s) of an object that must have none "
                               f"(-fno-slp-vectorize -fno-vectorize lost?): {worst}")
    # every object, the GEMM files included: no cross-lane read may take a register a packed fp32 op wrote just before it.
    # (The GEMM objects keep their packed conversions; what must not happen is the ADJACENCY, and that is checked here
    # instruction by instruction instead of trusted to the hand-placed fences.)
    hz = packed_crosslane_hazards(obj or _obj(src))
    if hz:
        raise RuntimeError(f"{base}: {len(hz)} cross-lane read(s) of a register written by a packed fp32 op within {HAZARD_WINDOW} "
                           f"instructions (DESIGN.md 4.1 hazard; put single_pass() / an empty asm on the operand): {hz[:4]}")
    if base not in NO_PACKED_FP32 and verbose:
        both = {f: v for f, v in rep.items() if v[1]}
        print(f"{base}: packed fp32 ops in {len(rep)} function(s), {len(both)} of them also read across lanes; "
              f"0 cross-lane reads fed by a packed op (checked per instruction)", flush=True)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    todo = [s for s in SOURCES if force or _stale(_obj(s), [s, os.path.abspath(__file__)] + HEADERS)]

    def compile_one(src):
        # the object only takes its final name once the ISA check has passed: a failed check (or a missing llvm-objdump)
        # must not leave an "up to date" object behind that the next build() would link unchecked
        tmp = _obj(src) + ".unchecked.o"
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", tmp]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return tmp

    if todo:
        if not os.path.exists(os.path.join(LLVM_BIN, "llvm-objdump")):
            raise RuntimeError(f"llvm-objdump not found under {LLVM_BIN} (set LLVM_BIN): the packed-fp32 ISA fence of the build "
                               "cannot run, refusing to produce unchecked objects")
        with ThreadPoolExecutor(max_workers=len(todo)) as ex:
            tmps = list(ex.map(compile_one, todo))
        try:
            for src, tmp in zip(todo, tmps):
                _check_isa(src, verbose, obj=tmp)
            for src, tmp in zip(todo, tmps):
                os.replace(tmp, _obj(src))
        finally:
            for tmp in tmps:
                if os.path.exists(tmp):
                    os.remove(tmp)
    objs = [_obj(s) for s in SOURCES]
    if todo or _stale(OUT, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
