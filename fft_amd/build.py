"""Build libarmfft_mi355.so (hipcc, gfx950 only) in-tree: fft_amd/lib/libarmfft_mi355.so.

Usage: python fft_amd/build.py [--force]   (by path: importing the package would load a stale or missing library)
The .so is git-ignored but travels to the GPU box with the tree (see .gpurunignore rules).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libarmfft_mi355.so")
OBJDIR = os.path.join(HERE, "build")
SOURCES = ["armfft_abi.cpp", "armfft_f32.hip", "armfft_fixed.hip", "armfft_large.hip", "armfft_gnss.hip", "armfft_resamp.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-ffp-contract=fast"]


def _deps():
    out = []
    for root, _d, files in os.walk(CSRC):
        out += [os.path.join(root, f) for f in files]
    out.append(os.path.join(HERE, "..", "include", "arm_fft_mi355.h"))
    return out


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    return hs + [os.path.join(HERE, "..", "include", "arm_fft_mi355.h")]


def _compile(src, force=False):
    """One object per source; recompiled only when the source or a header is newer than it."""
    obj = os.path.join(OBJDIR, src + ".o")
    if not force and not _stale(obj, [os.path.join(CSRC, src)] + _headers()):
        return obj
    cmd = [HIPCC] + FLAGS + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
    subprocess.run(cmd, check=True)
    return obj


def build(force=False, verbose=True):
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not force and not _stale(LIB, _deps()):
        if verbose:
            print("[fft_amd.build] %s is up to date, not rebuilt (--force rebuilds)" % os.path.relpath(LIB), flush=True)
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    if verbose:
        print("[fft_amd.build] hipcc gfx950:", " ".join(srcs), flush=True)
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(lambda f: _compile(f, force), srcs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
