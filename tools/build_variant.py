"""Developer tool: build libwost_hip.so with extra -D flags into elaina_amd/lib/variants/<name>/ (A/B experiments on the GPU box:
WOST_LIB=elaina_amd/lib/variants/<name>/libwost_hip.so python ...).  Usage: python tools/build_variant.py <name> "-DX=1 -DY" [unit.hip ...]
Units not listed are taken from the default build's objects (so only the listed ones are recompiled; default: all)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from elaina_amd import build as b  # noqa: E402

name, defs = sys.argv[1], sys.argv[2].split()
units = sys.argv[3:] or b.SOURCES
out = os.path.join(b.LIB_DIR, "variants", name)
os.makedirs(out, exist_ok=True)
b.build_library()
flags = [f for f in b.HIPCC_FLAGS if f != "-shared"] + defs
jobs, objs = [], []
for s in b.SOURCES:
    base = os.path.splitext(s)[0] + ".o"
    if s in units:
        obj = os.path.join(out, base)
        jobs.append(subprocess.Popen([b._hipcc()] + flags + ["-c", os.path.join(b.CSRC, s), "-o", obj]))
    else:
        obj = os.path.join(b.OBJ_DIR, base)
    objs.append(obj)
assert all(p.wait() == 0 for p in jobs)
lib = os.path.join(out, "libwost_hip.so")
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + objs + ["-o", lib])
print(lib)
