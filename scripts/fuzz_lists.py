#!/usr/bin/env python3
"""Randomised cross-check of the list builders under exact culling: supertile_sort.hip (masks prepared by preprocess, three
rectangle classes) against the global sort of the keys binning.hip emits -- same point_list, ranges and image, entry for entry.
python scripts/fuzz_lists.py [--seconds 120]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussian_transformer_amd import _lib, synth
from tests.helpers import oracle_scene
from tests.test_gpu_parity import _stage_dump

ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=120.0); a = ap.parse_args()
rng = np.random.default_rng(12345)
t0 = time.time(); n = 0
try:
    while time.time() - t0 < a.seconds:
        P = int(rng.choice([1, 7, 300, 4000, 9000, 30000, 90000]))
        W = int(rng.integers(16, 2000)); H = int(rng.integers(16, 1200))
        kw = dict(P=P, width=W, height=H, sh_degree=0, s0=float(10 ** rng.uniform(-2.3, -0.2)), seed=int(rng.integers(1 << 30)),
                  zmin=float(rng.choice([0.05, 1.0, 3.0])), zmax=float(rng.choice([3.0, 10.0, 200.0])))
        S = oracle_scene(synth.make_scene(**kw))
        d = {}
        for name, opts in (("sort", dict(two_level_sort=0, tile_lists=0, depth_buckets=0)), ("ss", dict(two_level_sort=1, tile_lists=2, depth_buckets=1))):
            for k, v in opts.items():
                _lib.set_option(k, v)
            d[name] = _stage_dump(S)
        x, y = d["sort"], d["ss"]
        assert x["n"] == y["n"], (kw, x["n"], y["n"])
        assert np.array_equal(x["ranges"], y["ranges"]), kw
        assert np.array_equal(x["point_list"], y["point_list"]), kw
        assert np.array_equal(x["color"], y["color"]), kw
        n += 1
        if n % 20 == 0:
            print(f"{n} scenes ok ({time.time() - t0:.0f} s)", flush=True)
finally:
    _lib.set_option("two_level_sort", 1); _lib.set_option("tile_lists", 2); _lib.set_option("depth_buckets", 1)
print(f"fuzz ok: {n} scenes, identical lists")
