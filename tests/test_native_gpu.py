"""GPU: a C++ host (no Python, no torch) links libsnag_hip.so through include/snag_hip.h and checks the scorer against a scalar
restatement of the reference formula -- the boundary really is the C ABI."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_calls_the_library(tmp_path):
    from snag_amd import _lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc) and shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    _lib.load()
    exe = str(tmp_path / "score_client")
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = [hipcc, "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "score_client.cpp"), "-o", exe,
           "-L", libdir, "-l:" + os.path.basename(_lib.LIB_PATH), f"-Wl,-rpath,{libdir}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-500:])
    assert "max abs err" in r.stdout
