"""-m gpu: a short optimisation loop through the whole HIP path with density control (scripts/train_synthetic.py):
render_fused -> fused L1+SSIM -> backward -> clone / split / prune on the device -> fused Adam.  The loss must fall and the
Gaussian set must change size, with every tensor (parameters, Adam moments, statistics) staying consistent."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_training_loop_with_density_control():
    path = os.path.join(os.path.dirname(__file__), "..", "scripts", "train_synthetic.py")
    spec = importlib.util.spec_from_file_location("train_synthetic", path)
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    out = mod.run(iters=240, P=6000, size=160, ncam=4, densify_from=60, densify_every=60)
    assert out["loss_last"] < 0.75 * out["loss_first"], out        # mean loss over all cameras, before / after
    assert out["psnr"] > out["psnr_first"] + 2.0, out
    assert out["P_end"] != out["P_start"], out
    assert any(h["event"] and (h["event"]["cloned"] + h["event"]["split"] + h["event"]["pruned"]) > 0 for h in out["history"]), out
