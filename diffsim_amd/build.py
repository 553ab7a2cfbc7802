"""Build libdiffsim_amd.so (gfx950) in-tree with hipcc.

``python -m diffsim_amd.build`` or ``diffsim_amd.build.build()``.  hipcc cross-compiles without
a GPU, so this runs in the build container; the resulting .so travels with the source tree.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libdiffsim_amd.so")
SOURCES = ["gemm.hip", "gemm_skinny.hip", "rowres.hip", "norm.hip", "attention.hip", "attn160.hip", "attention_fp8.hip", "pack.hip", "unet.hip", "vae.hip", "dit.hip"]
# the kernel sources written against the 16-bit type h16 are compiled a second time with h16 = fp16 (csrc/common.h): the
# compute dtype DSIM_F16, which the plain entry points forward to the *_f16 twins
F16_SOURCES = ["gemm.hip", "gemm_skinny.hip", "rowres.hip", "norm.hip", "attention.hip", "attn160.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "store.h"), os.path.join(HERE, "..", "include", "diffsim_amd.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# rowres.hip: the GELU runs beside MFMAs at one wave per SIMD, where hipcc's SLP-packed v_pk_*_f32 cost several times the two
# scalar instructions they replace (cdna_hip_programming.md, 4-wave attention pitfalls)
EXTRA_FLAGS = {"rowres.hip": ["-fno-slp-vectorize"]}


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(job) -> str:
    src, f16 = job
    obj = os.path.join(OBJ, src.replace(".hip", "_f16.o" if f16 else ".o"))
    path = os.path.join(CSRC, src)
    if _stale(obj, [path] + HEADERS):
        cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + (["-DDSIM_H16_IS_F16"] if f16 else []) + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    jobs = [(src, False) for src in SOURCES] + [(src, True) for src in F16_SOURCES]
    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(_compile, jobs))
    if _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print("built", LIB, os.path.getsize(LIB), "bytes")
    return LIB


if __name__ == "__main__":
    build(verbose=True)
    sys.exit(0)
