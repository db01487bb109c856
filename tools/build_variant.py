"""Build a VARIANT of libpolyhead.so for same-box A/B timing: recompile the named sources with extra -D flags and link them with
the other objects of the in-tree build -> tools/libpolyhead_<tag>.so (git-ignored, travels with gpurun); select it at run time
with PH_ALT_LIB=tools/libpolyhead_<tag>.so.
usage: python tools/build_variant.py <tag> "<flags>" ph_convup.hip [more.hip ...]"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from polyphonicformer_amd import build as B

tag, flags, srcs = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
B.build_library()                                  # the base objects must be current
tmp = os.path.join("/tmp", f"ph_variant_{tag}")
os.makedirs(tmp, exist_ok=True)
hipcc = B._hipcc()
objs = {os.path.basename(s)[:-4]: os.path.join(B.OBJ, os.path.basename(s)[:-4] + ".o") for s in B.sources()}
procs = []
for s in srcs:
    base = os.path.basename(s)[:-4]
    o = os.path.join(tmp, base + ".o")
    procs.append((base, o, subprocess.Popen([hipcc] + B.FLAGS + flags + ["-c", os.path.join(B.CSRC, base + ".hip"), "-o", o])))
for base, o, p in procs:
    if p.wait():
        raise SystemExit(f"hipcc failed on {base}")
    objs[base] = o
out = os.path.join(REPO, "tools", f"libpolyhead_{tag}.so")
subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + list(objs.values()) + ["-o", out])
print(out)
