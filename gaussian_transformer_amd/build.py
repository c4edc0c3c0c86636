"""Builds libgsr_hip.so (the C-ABI HIP library, include/gsr.h) in-tree with hipcc for gfx950.

    python -m gaussian_transformer_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the
repo snapshot.  No torch headers are involved: the library links only against libamdhip64.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libgsr_hip.so")
ARCH = "gfx950"

# translation unit -> extra flags
# -ffp-contract=off on the per-Gaussian stages: bit-parity of radii / tile rects / depth keys with
# the float32 CPU restatement (see csrc/gsr_device.h).  The compositing kernels keep FMA
# contraction and use native no-return float atomics.
SOURCES = {
    "gsr_api.hip": [],
    "preprocess.hip": ["-ffp-contract=off", "-fno-slp-vectorize"],
    "pergauss_bwd.hip": ["-ffp-contract=off", "-fno-slp-vectorize"],
    "binning.hip": ["-ffp-contract=off"],     # the tile-row span test must round exactly as in preprocess.hip
    "depth_order.hip": ["-ffp-contract=off"],
    "tile_lists.hip": ["-ffp-contract=off"],      # same tile-row spans as preprocess.hip, bit for bit
    "supertile_sort.hip": ["-ffp-contract=off"],  # likewise
    # -fno-slp-vectorize: the SLP pass packs pairs of scalar f32 ops into v_pk_* and pays for it in v_mov shuffles and
    # registers (bwd: 86 -> 71 VGPRs, 5 -> 7 waves/SIMD; fwd 178 -> 157 us, bwd 387 -> 352 us at config 3, measured)
    "composite_fwd.hip": ["-fno-slp-vectorize"],
    "composite_bwd.hip": ["-munsafe-fp-atomics", "-fno-slp-vectorize"],
    "ssim_loss.hip": [],
    "knn.hip": [],
    "adam.hip": [],
}
COMMON = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-result",
          "-fgpu-rdc" if False else "-fno-gpu-rdc"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built on this machine")
    return exe


def _newer(src_list, target) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build_variant(tag: str, defines, verbose: bool = False, force: bool = False) -> str:
    """Variant build: every TU recompiled with extra -D flags into libgsr_hip_<tag>.so (ablations; "dbg" = the capacity-assert build,
    -DGSR_DEBUG_BOUNDS, csrc/gsr_internal.h).  Loaded instead of the product library through GSR_LIB_PATH."""
    cc = hipcc()
    out = os.path.join(HERE, f"libgsr_hip_{tag}.so")
    odir = os.path.join(OBJ, tag)
    os.makedirs(odir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")] + [os.path.abspath(__file__)]
    headers += [os.path.join(HERE, "..", "include", h) for h in ("gsr.h", "gsr_loss.h", "gsr_knn.h", "gsr_optim.h")]
    jobs, objs = [], []
    for src, extra in SOURCES.items():
        s, o = os.path.join(CSRC, src), os.path.join(odir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer([s] + headers, o):
            jobs.append([cc, "-c", s, "-o", o] + COMMON + extra + [f"-D{d}" for d in defines])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _newer(objs, out):
        subprocess.check_call([cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", out] + objs)
    return out


def build_debug_bounds(force: bool = False, verbose: bool = False) -> str:
    return build_variant("dbg", ["GSR_DEBUG_BOUNDS"], verbose=verbose, force=force)


def build_hip(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "gsr.h"))
    headers.append(os.path.join(HERE, "..", "include", "gsr_loss.h"))
    headers.append(os.path.join(HERE, "..", "include", "gsr_knn.h"))
    headers.append(os.path.join(HERE, "..", "include", "gsr_optim.h"))
    headers.append(os.path.abspath(__file__))
    cc = hipcc()
    jobs = []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _newer([s] + headers, o):
            jobs.append((src, [cc, "-c", s, "-o", o] + COMMON + extra))

    def run(job):
        name, cmd = job
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {name}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return name

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _newer(objs, LIB):
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))
    if "--debug-bounds" in sys.argv:
        print(build_debug_bounds(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))
