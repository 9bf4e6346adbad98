"""Builds libpnr_hip.so (the C-ABI HIP library, gfx950 only) with hipcc.

`python -m pointnerf2studio_amd.build` or `build_library()`; hipcc cross-compiles for gfx950
without a GPU.  In a source tree the library is built next to the package (in-tree, so it travels with the tree);
an INSTALLED copy (pip install: HIP sources and the header ship as package data, pyproject.toml) that holds no
library builds it on first use -- `_lib.load()` -- into the package directory, or into a per-user cache when that
is read-only, and fails loudly where there is no hipcc (no CPU fallback exists or is wanted).
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
# the C-ABI header: include/pnr.h of the source tree, or the copy setup.py puts into an installed package
INCLUDE = os.path.join(ROOT, "include")
if not os.path.exists(os.path.join(INCLUDE, "pnr.h")):
    INCLUDE = os.path.join(PKG_DIR, "include")
LIB_NAME = "libpnr_hip.so"
LIB_PATH = os.path.join(PKG_DIR, LIB_NAME)
OBJ_DIR = os.path.join(PKG_DIR, "csrc", "_obj")

SOURCES = ["pnr_scan.hip", "pnr_scene.hip", "pnr_query.hip", "pnr_shade.hip", "pnr_shade_fp32.hip", "pnr_shade_bf16.hip",
           "pnr_render.hip", "pnr_train.hip", "pnr_train_chain.hip", "pnr_optim.hip"]

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


def find_hipcc():
    exe = shutil.which("hipcc") or os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")
    return exe if os.path.exists(exe) else None


def hipcc() -> str:
    exe = find_hipcc()
    if exe is None:
        raise RuntimeError("hipcc not found; a ROCm toolchain is required to build libpnr_hip.so")
    return exe


def sources_present() -> bool:
    return (all(os.path.exists(os.path.join(CSRC, s)) for s in SOURCES)
            and os.path.exists(os.path.join(INCLUDE, "pnr.h")))


def _writable(d: str) -> bool:
    try:
        os.makedirs(d, exist_ok=True)
        probe = os.path.join(d, f".pnr_write_probe_{os.getpid()}")
        with open(probe, "w"):
            pass
        os.remove(probe)
        return True
    except OSError:
        return False


def cache_dir() -> str:
    """Where an installed copy in a read-only location keeps its library: $PNR_CACHE_DIR, else
    ~/.cache/pointnerf2studio_amd/<hash of the sources>: a new release never loads an old build."""
    import hashlib
    h = hashlib.sha256()
    for s in sorted(SOURCES) + ["pnr_internal.h", "pnr_shade_common.h", "pnr_train_chain.h"]:
        p = os.path.join(CSRC, s)
        if os.path.exists(p):
            h.update(open(p, "rb").read())
    base = os.environ.get("PNR_CACHE_DIR") or os.path.join(os.path.expanduser("~"), ".cache", "pointnerf2studio_amd")
    return os.path.join(base, h.hexdigest()[:16])


def default_out_dir() -> str:
    return PKG_DIR if _writable(PKG_DIR) else cache_dir()


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False, out_dir: str = None) -> str:
    """Compiles the nine .hip files and links libpnr_hip.so into `out_dir` (default: the package directory).  Returns
    the library's path."""
    lib_path = LIB_PATH if out_dir is None else os.path.join(out_dir, LIB_NAME)
    obj_dir = OBJ_DIR if out_dir is None else os.path.join(out_dir, "_obj")
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.join(CSRC, "pnr_internal.h"), os.path.join(CSRC, "pnr_shade_common.h"),
               os.path.join(CSRC, "pnr_train_chain.h"),
               os.path.join(INCLUDE, "pnr.h"), os.path.abspath(__file__)]
    cc = hipcc()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
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
    objs = [os.path.join(obj_dir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(lib_path, objs):
        # (linked under a temporary name and renamed: a second process never loads a half-written library)
        tmp = f"{lib_path}.{os.getpid()}.tmp"
        run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs])
        os.replace(tmp, lib_path)
    return lib_path


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
