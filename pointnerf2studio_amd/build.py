"""Builds libpnr_hip.so (the C-ABI HIP library, gfx950 only) in-tree with hipcc.

`python -m pointnerf2studio_amd.build` or `build_library()`; hipcc cross-compiles for gfx950
without a GPU.  The library is NOT built lazily at import on a GPU box: `_lib.load()` fails
loudly when it is missing (no CPU fallback exists or is wanted).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB_NAME = "libpnr_hip.so"
LIB_PATH = os.path.join(PKG_DIR, LIB_NAME)
OBJ_DIR = os.path.join(PKG_DIR, "csrc", "_obj")

SOURCES = ["pnr_scan.hip", "pnr_scene.hip", "pnr_query.hip", "pnr_shade.hip", "pnr_shade_fp32.hip", "pnr_shade_bf16.hip",
           "pnr_render.hip", "pnr_train.hip", "pnr_train_chain.hip"]

# -ffp-contract=off: the voxel coordinate, the sample position (o + d*t) and the neighbour distance
# must be evaluated exactly as the reference / oracle do (no FMA contraction); the MLP runs on fp32
# MFMA, whose fma chain is explicit.
# -pragma-unroll-threshold: the pair kernels are ONE basic block of 3360 MFMAs with everything else placed between them by
# hand; `#pragma unroll` silently gives up above LLVM's default of 16 k instructions per unrolled loop.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-mllvm", "-pragma-unroll-threshold=4000000",
         "-Wall", "-Wno-unused-function", "-I", INCLUDE, "-I", CSRC]


# per-file additions.  pnr_shade_fp32.hip: MFMA results in VGPRs rather than AGPRs -- every output value of the MLP is
# read by vector-ALU code (LeakyReLU) right away, and one wave per SIMD does not overlap its own VALU and MFMA work
# (tools/ub_mfma_dep.hip): the 512 v_accvgpr_read per tile that the AGPR form needs cost their full issue time.
FILE_FLAGS = {"pnr_shade_fp32.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
              "pnr_train_chain.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; a ROCm toolchain is required to build libpnr_hip.so")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, "pnr_internal.h"), os.path.join(CSRC, "pnr_shade_common.h"),
               os.path.join(CSRC, "pnr_train_chain.h"),
               os.path.join(INCLUDE, "pnr.h"), os.path.abspath(__file__)]
    cc = hipcc()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append([cc, *FLAGS, *FILE_FLAGS.get(src, []), "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0 and "Unknown command line argument" in p.stderr:
            # a toolchain without one of the FILE_FLAGS options (they are tuning, not semantics): build without them
            extra = {f for fl in FILE_FLAGS.values() for f in fl if f != "-mllvm"}
            slim = [c for i, c in enumerate(cmd) if c not in extra and not (c == "-mllvm" and cmd[i + 1] in extra)]
            print(f"note: {os.path.basename(cmd[-3])}: per-file flags not supported by this hipcc, building without",
                  file=sys.stderr)
            p = subprocess.run(slim, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{p.stdout}\n{p.stderr}")
        if verbose and p.stderr.strip():
            print(p.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(OBJ_DIR, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB_PATH, objs):
        run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objs])
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
