"""Build build/variants/<name>.so: the library with ONE source recompiled under extra -D flags (kernel A/B builds for
tools/*_ab.py).  usage: build_variant.py <name> <source.hip> [-DFLAG ...] [--from FILE]   (the other objects come from build/obj; --from: compile
FILE, e.g. an older revision of the source written out by git show, in place of csrc/<source.hip>)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd.csrc import build as B

name, src, defs = sys.argv[1], sys.argv[2], sys.argv[3:]
alt = None
if "--from" in defs:
    i = defs.index("--from")
    alt = defs[i + 1]
    defs = defs[:i] + defs[i + 2:]
B.build(verbose=False)                                    # the unchanged objects
src_path = os.path.join(B.HERE, os.path.basename(src))
vdir = os.path.join(B.ROOT, "build", "variants")
os.makedirs(vdir, exist_ok=True)
obj = os.path.join(vdir, f"{name}.{os.path.basename(src)}.o")
cmd = ["/opt/rocm/bin/hipcc"] + B.FLAGS + B.EXTRA_FLAGS.get(os.path.basename(src), []) + defs + ["-c", alt or src_path, "-o", obj]
subprocess.check_call(cmd)
B._check_isa(src_path, False, obj=obj)
objs = [obj if s == src_path else B._obj(s) for s in B.SOURCES]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(vdir, f"{name}.so")] + objs)
print(os.path.join(vdir, f"{name}.so"))
