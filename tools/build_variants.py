#!/usr/bin/env python3
"""Developer tool: build alternative libwost_hip.so variants (extra -D definitions on chosen translation
units) into elaina_amd/lib/variants/<name>.so; select one at run time with WOST_LIB=<path>.
Usage: build_variants.py name=unit.hip[,unit2.hip]:-DX=1,-DY ...   (units default to wost_hip.hip)
The normal library (elaina_amd/lib/libwost_hip.so) is never touched."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elaina_amd import build as B  # noqa: E402

B.build_library()
vdir = os.path.join(B.LIB_DIR, "variants")
os.makedirs(vdir, exist_ok=True)
flags = [f for f in B.HIPCC_FLAGS if f != "-shared"]
jobs = []
for spec in sys.argv[1:]:
    name, rest = spec.split("=", 1)
    units, defs = (rest.split(":", 1) + [""])[:2] if ":" in rest else ("wost_hip.hip", rest)
    units = units.split(",")
    defs = [d for d in defs.split(",") if d]
    objs = []
    for s in B.SOURCES:
        base = os.path.splitext(s)[0]
        if s in units:
            obj = os.path.join(vdir, "%s_%s.o" % (name, base))
            jobs.append((name, subprocess.Popen([B._hipcc()] + flags + defs + ["-c", os.path.join(B.CSRC, s), "-o", obj])))
        else:
            obj = os.path.join(B.OBJ_DIR, base + ".o")
        objs.append(obj)
    jobs.append((name, objs))
pending = {}
for name, j in jobs:
    if isinstance(j, subprocess.Popen):
        if j.wait() != 0:
            raise SystemExit("compile failed for variant " + name)
    else:
        out = os.path.join(vdir, name + ".so")
        subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + j + ["-o", out])
        print(out)
