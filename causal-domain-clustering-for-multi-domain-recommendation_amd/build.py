"""Builds csrc/*.hip into the in-tree C-ABI library libcdcmdr.so (gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting
.so travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import hashlib
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), "include")
LIB_PATH = os.path.join(PKG_DIR, "libcdcmdr.so")
STAMP = os.path.join(PKG_DIR, "libcdcmdr.so.stamp")

SOURCES = ["misc.hip", "embedding.hip", "gemm.hip", "gemm2.hip", "rowops.hip", "cgc.hip", "pair.hip", "head.hip", "tower.hip", "metrics.hip", "attention.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment", "-ffp-contract=off"]
FLAGS += os.environ.get("CDC_EXTRA_HIPCC_FLAGS", "").split()      # probe builds (e.g. -DTW_TRACE=1: tools/tower_trace.py)


def _digest():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)) + ["../../include/cdcmdr.h"]:
        path = os.path.normpath(os.path.join(CSRC, name))
        if os.path.isfile(path):
            h.update(name.encode())
            with open(path, "rb") as f:
                h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def is_fresh():
    if not (os.path.exists(LIB_PATH) and os.path.exists(STAMP)):
        return False
    with open(STAMP) as f:
        return f.read().strip() == _digest()


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link libcdcmdr.so. Returns the library path."""
    if not force and is_fresh():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    objs = []
    procs = []
    build_dir = os.path.join(PKG_DIR, "build")
    os.makedirs(build_dir, exist_ok=True)
    # an object is kept while its source, the headers and the flags are unchanged (a one-file edit recompiles one file)
    hh = hashlib.sha256(" ".join(FLAGS).encode())
    for name in sorted(os.listdir(CSRC)) + ["../../include/cdcmdr.h"]:
        if name.endswith(".h"):
            with open(os.path.normpath(os.path.join(CSRC, name)), "rb") as f:
                hh.update(f.read())
    for src in SOURCES:
        obj = os.path.join(build_dir, src.replace(".hip", ".o"))
        objs.append(obj)
        with open(os.path.join(CSRC, src), "rb") as f:
            want = hashlib.sha256(hh.digest() + f.read()).hexdigest()
        stamp = obj + ".stamp"
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == want:
            continue
        cmd = [hipcc] + FLAGS + ["-I", INCLUDE, "-I", CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[cdcmdr build]", " ".join(cmd), flush=True)
        procs.append((src, stamp, want, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, stamp, want, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        with open(stamp, "w") as f:
            f.write(want)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print("[cdcmdr build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    with open(STAMP, "w") as f:
        f.write(_digest())
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
