"""Build libgsasr_splat.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m gsasr_amd.build            # or gsasr_amd.build.build_library()

The library has no torch / python dependency: it is the C ABI of include/gsasr_splat.h.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SRC = os.path.join(PKG, "csrc", "gsasr_splat.hip")
INC = os.path.join(ROOT, "include")
LIB_DIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIB_DIR, "libgsasr_splat.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics", "-fno-slp-vectorize", "-fvisibility=hidden",
               "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm's hipcc to build libgsasr_splat.so)")
    return exe


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [SRC, os.path.join(INC, "gsasr_splat.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc(), *HIPCC_FLAGS, "-I", INC, SRC, "-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


AUTOGRAD_SRC = os.path.join(PKG, "csrc", "gsasr_autograd.cpp")
AUTOGRAD_EXT = os.path.join(LIB_DIR, "_gsasr_autograd.so")


def build_autograd_node(force: bool = False, verbose: bool = False) -> str:
    """gsasr_amd/lib/_gsasr_autograd.so: the drop-in GSCUDA node as a C++ torch::autograd::Function (host code only: g++
    against this interpreter's torch; gsasr_amd/_cpp_node.py loads it when present, the Python Functions serve otherwise)."""
    if not force and os.path.exists(AUTOGRAD_EXT) and \
            os.path.getmtime(AUTOGRAD_EXT) >= max(os.path.getmtime(AUTOGRAD_SRC), os.path.getmtime(os.path.join(INC, "gsasr_splat.h"))):
        return AUTOGRAD_EXT
    import sysconfig

    import torch
    from torch.utils import cpp_extension as ce
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        raise RuntimeError("g++ not found")
    os.makedirs(LIB_DIR, exist_ok=True)
    libdirs = ce.library_paths()
    cmd = [cxx, "-O2", "-std=c++17", "-shared", "-fPIC", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           "-DTORCH_EXTENSION_NAME=_gsasr_autograd", "-DTORCH_API_INCLUDE_EXTENSION_H", AUTOGRAD_SRC, f"-I{INC}",
           *[f"-I{p}" for p in ce.include_paths()], f"-I{sysconfig.get_paths()['include']}",
           *[f"-L{p}" for p in libdirs], "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python",
           *[f"-Wl,-rpath,{p}" for p in libdirs], "-o", AUTOGRAD_EXT]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return AUTOGRAD_EXT


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
    print(build_autograd_node(force="--force" in sys.argv, verbose=True))
