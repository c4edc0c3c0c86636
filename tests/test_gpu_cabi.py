"""-m gpu: a plain C program (tests/cabi/c_client.c) drives libgsr_hip.so directly -- no Python, no
torch, no C++ types across the boundary -- including the error paths (codes, not aborts)."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_client_renders_and_differentiates(tmp_path):
    pkg = os.path.join(ROOT, "gaussian_transformer_amd")
    exe = str(tmp_path / "c_client")
    cc = shutil.which("gcc") or "gcc"
    cmd = [cc, "-std=c11", "-O1", os.path.join(ROOT, "tests", "cabi", "c_client.c"), "-I", os.path.join(ROOT, "include"),
           "-I/opt/rocm/include", "-L", pkg, "-lgsr_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
           f"-Wl,-rpath,{pkg}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "C client ok" in r.stdout
