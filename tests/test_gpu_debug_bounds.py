"""-m gpu: the capacity-assert build (libgsr_hip_dbg.so, -DGSR_DEBUG_BOUNDS: csrc/gsr_internal.h GSR_IDX_OK) over random scenes, the two
dataset-derived configs and the segmented reverse pass: no index reaches the capacity a host-side plan gave it.  The library is loaded
once per process, so the debug build runs in a child process (GSR_LIB_PATH)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DBG = os.path.join(ROOT, "gaussian_transformer_amd", "libgsr_hip_dbg.so")


def test_product_build_has_the_checks_compiled_out():
    import torch
    from gaussian_transformer_amd import _lib
    from gaussian_transformer_amd.rasterizer import get_backend
    be = get_backend()
    _, ib, _ = be._sizes(1, 64, 48)
    img = torch.zeros(ib, dtype=torch.uint8, device="cuda")
    out = np.zeros(12, np.uint32)
    _lib.check(be.lib.gsr_debug_read_bound_errors(torch.cuda.current_stream().cuda_stream, 0, None, 64, 48, img.data_ptr(), out.ctypes.data), "read")
    assert out[8] == 0 and out[9] == 0xdead and not out[:8].any()


@pytest.mark.timeout(900)
def test_debug_build_finds_no_capacity_violation():
    assert os.path.exists(DBG), "libgsr_hip_dbg.so missing: __graft_entry__.build() / python -m gaussian_transformer_amd.build --debug-bounds"
    env = dict(os.environ, GSR_LIB_PATH=DBG)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "debug_bounds_run.py"), "--scenes", "32"], env=env, capture_output=True, text=True,
                       timeout=850)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert line, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    rep = json.loads(line[-1])
    assert rep["debug_build"] == 1 and rep["selftest"] == [99, 5, 4], rep         # the checks are in and they do trip
    assert rep["scenes"] >= 38 and not rep["violations"] and r.returncode == 0, rep
