"""The C ABI used from plain C (no Python, no torch in the process): builds tests/c_abi/c_abi_check.c with
gcc against libgsasr_splat.so + the oracle library and runs it on the GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_c_program_links_and_matches_oracle(tmp_path):
    from gsasr_amd import _cabi
    from oracle import gs_oracle
    lib = _cabi.LIB_PATH
    ref = gs_oracle.build()
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cc = shutil.which("gcc") or shutil.which("cc")
    assert cc and os.path.exists(lib) and os.path.exists(ref)
    exe = str(tmp_path / "c_abi_check")
    cmd = [cc, "-O1", "-std=c11", "-D__HIP_PLATFORM_AMD__", f"-I{rocm}/include", f"-I{ROOT}/include",
           os.path.join(ROOT, "tests", "c_abi", "c_abi_check.c"), lib, ref, f"-L{rocm}/lib", "-lamdhip64", "-lm",
           f"-Wl,-rpath,{os.path.dirname(lib)}", f"-Wl,-rpath,{os.path.dirname(ref)}", f"-Wl,-rpath,{rocm}/lib", "-o", exe]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "C-ABI CHECK OK" in out.stdout, out.stdout + out.stderr
