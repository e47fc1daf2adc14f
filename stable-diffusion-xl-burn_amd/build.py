"""Builds libsdxl_mi355.so (HIP kernels + C++ engine + C ABI) for gfx950 with hipcc, in-tree.

    python stable-diffusion-xl-burn_amd/build.py [--force]

hipcc cross-compiles gfx950 without a GPU, so this runs in the CPU-only build container; the resulting
.so travels with the repo snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsdxl_mi355.so")
ARCH = "gfx950"
SOURCES = ["igemm.hip", "igemm_glds.hip", "igemm_wreg.hip", "igemm_measure.hip", "norm.hip", "attention.hip", "elementwise.hip", "capi.hip",
           "specs.cpp", "weights.cpp", "unet.cpp", "vae.cpp", "sampler.cpp", "clip.cpp", "comm.cpp"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _measure() -> bool:
    return os.environ.get("SDXL_MEASURE") == "1"


def _paths():
    """(object dir, library): the measurement build lives NEXT to the release library (lib/libsdxl_mi355_measure.so, loaded with
    SDXL_MEASURE_LIB=1), so both travel to the GPU box and a sweep never replaces the library the tests / bench.py run"""
    return (OBJ + "_measure", LIB.replace(".so", "_measure.so")) if _measure() else (OBJ, LIB)


def _flags() -> list:
    # SDXL_MEASURE=1 (or `build.py --measure`): also build the A/B partners, dead-end variants and measurement-only modes of
    # the GEMM / attention kernels plus the semantics-changing debug knobs; the default (release) library has none of them
    return FLAGS + (["-DSDXL_MEASURE"] if _measure() else [])


def _deps_hash(src: str) -> str:
    h = hashlib.sha256()
    for f in [src] + [os.path.join(CSRC, x) for x in ("kernels.h", "engine.h", "igemm_common.h")] + \
            [os.path.join(HERE, "..", "include", "sdxl_mi355.h")]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(_flags()).encode())
    return h.hexdigest()


def _compile(name: str, force: bool) -> str:
    src = os.path.join(CSRC, name)
    obj = os.path.join(_paths()[0], name + ".o")
    stamp = obj + ".sha"
    want = _deps_hash(src)
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want:
        return obj
    cmd = [_hipcc()] + _flags() + (["-x", "hip"] if name.endswith(".hip") else []) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {name}:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as fh:
        fh.write(want)
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    obj_dir, lib = _paths()
    os.makedirs(obj_dir, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(lambda n: _compile(n, force), SOURCES))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(lib) or os.path.getmtime(lib) < newest:
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {lib} ({os.path.getsize(lib) / 1e6:.1f} MB)")
    return lib


if __name__ == "__main__":
    if "--measure" in sys.argv:
        os.environ["SDXL_MEASURE"] = "1"
    build(force="--force" in sys.argv)
