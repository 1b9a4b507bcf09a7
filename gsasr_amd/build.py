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
CSRC = os.path.join(PKG, "csrc")
# the translation units of the library (csrc/gsasr_splat.hip is the same code as ONE unit: the micro-benchmark tools/mb.hip builds that)
PARTS = ["splat_api", "splat_plan", "splat_forward", "splat_backward", "splat_backward_home", "splat_step", "splat_sampled", "splat_shard"]
INC = os.path.join(ROOT, "include")
LIB_DIR = os.path.join(PKG, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB = os.path.join(LIB_DIR, "libgsasr_splat.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-slp-vectorize", "-fvisibility=hidden",
               "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm's hipcc to build libgsasr_splat.so)")
    return exe


def _deps():
    return [os.path.join(CSRC, p + ".hip") for p in PARTS] + [os.path.join(CSRC, "splat_common.h"), os.path.join(CSRC, "splat_bwd_sweep.h"), os.path.join(INC, "gsasr_splat.h")]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def build_library(force: bool = False, verbose: bool = False, extra_flags=(), out: str = LIB) -> str:
    """compile the translation units in parallel (hipcc, gfx950) and link them into one shared library"""
    if not force and out == LIB and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    cc = hipcc()
    common = max(os.path.getmtime(os.path.join(CSRC, "splat_common.h")), os.path.getmtime(os.path.join(CSRC, "splat_bwd_sweep.h")))
    header = os.path.getmtime(os.path.join(INC, "gsasr_splat.h"))
    # objects are cached per output name AND per flag list (a what-if build with other -D switches must not link stale objects)
    import hashlib
    flag_tag = hashlib.sha1(" ".join([*HIPCC_FLAGS, *extra_flags]).encode()).hexdigest()[:8]
    tag = ("" if out == LIB else "_" + os.path.splitext(os.path.basename(out))[0]) + "_" + flag_tag

    def one(part):
        src, obj = os.path.join(CSRC, part + ".hip"), os.path.join(OBJ_DIR, part + tag + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), common, header):
            return obj
        tmp = f"{obj}.{os.getpid()}.tmp"      # (ranks that build at first import must not link each other's half-written objects)
        cmd = [cc, *HIPCC_FLAGS, *extra_flags, "-I", INC, "-c", src, "-o", tmp]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        os.replace(tmp, obj)
        return obj

    with ThreadPoolExecutor(max_workers=len(PARTS)) as ex:
        objs = list(ex.map(one, PARTS))
    tmp_out = f"{out}.{os.getpid()}.tmp"
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden", *objs, "-o", tmp_out]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(tmp_out, out)
    return out


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
