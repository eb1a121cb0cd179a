"""Build libtactilesr_hip.so in-tree with hipcc for gfx950 (no torch linkage).

    python -m tactilesr_amd.build          # incremental
    python -m tactilesr_amd.build --force

The .so is git-ignored but travels with the gpurun snapshot; nothing is JIT-built at
import time on the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libtactilesr_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-I", CSRC, "-I", os.path.join(os.path.dirname(HERE), "include")]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def source_hash() -> str:
    """sha256[:16] over every kernel source (csrc/*.hip, csrc/*.h, include/*.h): bench.py compares it with the hash a
    committed PMC profile was taken at, so `roofline.traffic` is never quoted for kernels that have changed since."""
    import hashlib
    inc = os.path.join(os.path.dirname(HERE), "include")
    h = hashlib.sha256()
    for d in (CSRC, inc):
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".h")):
                h.update(f.encode())
                h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def source_hashes() -> dict:
    """{file name: sha256[:16]} of every kernel source and header: a PMC profile records them, and bench.py quotes a
    kernel's measured HBM traffic only while the file that defines the kernel and every shared header are unchanged."""
    import hashlib
    inc = os.path.join(os.path.dirname(HERE), "include")
    out = {}
    for d in (CSRC, inc):
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".h")):
                out[f] = hashlib.sha256(open(os.path.join(d, f), "rb").read()).hexdigest()[:16]
    return out


def kernel_source_state(kernel_name: str, hashes: dict = None) -> dict:
    """The subset of `source_hashes()` a kernel's code depends on: the .hip file that defines it + the transitive closure
    of the csrc headers that file includes (the public include/tactilesr_hip.h holds declarations and two host-facing
    structs; a new entry point there does not change a kernel)."""
    import re
    h = source_hashes() if hashes is None else hashes
    base = kernel_name.split("<")[0].split("(")[0].replace("void ", "").strip()
    dep, todo = {}, []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hip") and base and base in open(os.path.join(CSRC, f)).read():
            todo.append(f)
    while todo:
        f = todo.pop()
        if f in dep or f == "tactilesr_hip.h" or not os.path.exists(os.path.join(CSRC, f)):
            continue
        dep[f] = h.get(f)
        todo += re.findall(r'#include\s+"([^"]+)"', open(os.path.join(CSRC, f)).read())
    return dep


def _newer(path, deps):
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    inc = os.path.join(os.path.dirname(HERE), "include")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]   # tsr_conv_desc lives there
    jobs = []
    objs = []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src[:-4] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + headers):
            jobs.append([HIPCC, *FLAGS, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print("[build]", " ".join(cmd[-4:]), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _newer(LIB, objs):
        run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB])
    return LIB


if __name__ == "__main__":
    if "--source-hash" in sys.argv:
        print(source_hash())
    else:
        print(build(force="--force" in sys.argv))
