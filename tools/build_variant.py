#!/usr/bin/env python3
"""Build an experimental variant of libtactilesr_hip.so with extra -D flags:

    python tools/build_variant.py NAME -DTSR_EXP_FOO [-D...]

-> tactilesr_amd/lib/exp/NAME/libtactilesr_hip.so (select it with TSR_LIB_OVERRIDE=<that path> TSR_ALLOW_VARIANT=1: a
variant library reports itself through tsr_build_flags() and the binding refuses it otherwise).
Only the sources that mention a given macro are recompiled; the other objects are reused.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tactilesr_amd import build as B  # noqa: E402

name, defs = sys.argv[1], sys.argv[2:] + ["-DTSR_VARIANT_BUILD=1"]       # api.hip: tsr_build_flags() reports the variant
B.build(verbose=False)
out = os.path.join(B.LIBDIR, "exp", name)
os.makedirs(out, exist_ok=True)
macros = [d[2:].split("=")[0] for d in defs]
hdr_hit = any(m in open(os.path.join(B.CSRC, h)).read() for h in os.listdir(B.CSRC) if h.endswith(".h") for m in macros)
objs, jobs = [], []
for src in B._sources():
    s = os.path.join(B.CSRC, src)
    if hdr_hit or any(m in open(s).read() for m in macros):
        o = os.path.join(out, src[:-4] + ".o")
        jobs.append([B.HIPCC, *B.FLAGS, *defs, "-c", s, "-o", o])
    else:
        o = os.path.join(B.OBJDIR, src[:-4] + ".o")
    objs.append(o)


def run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(r.stderr)


with ThreadPoolExecutor(max_workers=4) as ex:
    list(ex.map(run, jobs))
lib = os.path.join(out, "libtactilesr_hip.so")
run([B.HIPCC, f"--offload-arch={B.ARCH}", "-shared", "-fPIC", *objs, "-o", lib])
print(lib)
