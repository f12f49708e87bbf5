"""A second BUILD of the product library with extra -D flags on one source (for tools/attn_lib_ab.py / gemm_lib_ab.py: two builds interleaved in one process):
    python tools/build_variant.py NAME SOURCE.hip -DFLAG=1 [-DFLAG2=...]   ->  tools/probe/bin/libunigen_NAME.so
The other objects are the product's own (unigen_amd/csrc/*.o, built first if stale)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unigen_amd import build as Bd

name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
Bd.build()
out = os.path.join(ROOT, "tools", "probe", "bin"); os.makedirs(out, exist_ok=True)
obj = os.path.join(out, f"{src.replace('.hip', '')}_{name}.o")
subprocess.run([Bd.HIPCC, *flags, *Bd.FLAGS, *Bd.EXTRA.get(src, []), "-c", os.path.join(Bd.CSRC, src), "-o", obj], check=True)
objs = [obj if s_ == src else os.path.join(Bd.CSRC, s_.replace(".hip", ".o")) for s_ in Bd.SOURCES]
lib = os.path.join(out, f"libunigen_{name}.so")
subprocess.run([Bd.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], check=True)
print(lib)
