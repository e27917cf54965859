"""Builds libivit.so (the C-ABI engine, include/ivit.h) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU.  The shared object lands next to this file so that it travels
with the source snapshot to the GPU box; it is git-ignored.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_build")
LIB_PATH = os.path.join(HERE, "libivit.so")
SOURCES = ["engine.hip", "kernels_gemm.hip", "kernels_attn.hip", "kernels_misc.hip"]
HEADERS = ["common.h", "kernels.h", "gemm_kernel.h", "gemm256_kernel.h", "gemm256s_kernel.h",  os.path.join("..", "..", "include", "ivit.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _digest() -> str:
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    stamp = os.path.join(OUT_DIR, "stamp")
    digest = _digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(stamp) and open(stamp).read() == digest:
        return LIB_PATH
    cc = _hipcc()

    def compile_one(src: str) -> str:
        obj = os.path.join(OUT_DIR, os.path.splitext(src)[0] + ".o")
        cmd = [cc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    link = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH, *objs,
            "-Wl,-rpath,/opt/rocm/lib", "-Wl,-soname,libivit.so"]
    r = subprocess.run(link, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(digest)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
