#!/usr/bin/env python
"""Wave timeline of the two compositing kernels on a BASELINE config (instrumented kernels, gsr_debug_read_wave_trace).

    python scripts/wave_trace.py [--config cfg2_table_300k_800 ...] [--opt name=value ...]

Prints, per direction: kernel span, how many waves ran, the distribution of wave durations, the time per staged entry
of the longest waves, and how many waves were running at 25/50/75/90 % of the span (a latency-bound tail shows as a
handful of waves alive for most of the kernel).
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", nargs="*", default=["cfg2_table_300k_800"])
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--dump", default="")
    args = ap.parse_args()
    import torch
    from gaussian_transformer_amd import _lib, synth
    from scripts.lane_counters import count_once
    dev = torch.device("cuda", 0)
    opts = [(kv.split("=")[0], int(kv.split("=")[1])) for kv in args.opt]
    lib = _lib.load()
    TICK_US = 1e3 / torch.cuda.get_device_properties(0).clock_rate if False else 0.01      # wall_clock64: 100 MHz constant-rate counter
    for cfg in args.config:
        sc = synth.make_config(cfg, seed=0)
        count_once(sc, dev, opts)           # warm-up (allocations, first-use set-up); also the lane-slot counters
        cnt = count_once(sc, dev, opts)
        for which in (0, 1):                # reading clears the timeline: the counting runs above filed their waves too (other ids)
            junk = np.zeros((1 << 20, 4), dtype=np.uint32)
            _lib.check(lib.gsr_debug_read_wave_trace(which, junk.ctypes.data_as(C.c_void_p), 1 << 20), "clear trace")
        lib.gsr_set_profiling(1)
        count_once(sc, dev, opts, mode=2)   # wave timeline only: the kernels run at their normal speed
        lib.gsr_set_profiling(0)
        names = (C.c_char_p * _lib.GSR_NUM_STAGES)(); ms = (C.c_float * _lib.GSR_NUM_STAGES)()
        lib.gsr_get_stage_times(names, ms)
        stage = {names[i].decode(): float(ms[i]) for i in range(_lib.GSR_NUM_STAGES)}
        units = 1 << 20
        for which, tag in ((0, "fwd"), (1, "bwd")):
            buf = np.zeros((units, 4), dtype=np.uint32)
            _lib.check(lib.gsr_debug_read_wave_trace(which, buf.ctypes.data_as(C.c_void_p), units), "read trace")
            ran = buf[:, 1] != 0
            b = buf[ran].astype(np.int64)
            hw = b[:, 3] >> 12
            b[:, 3] &= 0xfff
            xcc, se, cu, simd = (hw >> 16) & 15, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3
            simd_key = ((xcc * 8 + se) * 16 + cu) * 4 + simd
            if not len(b):
                continue
            t0 = b[:, 0].min()
            st, en = (b[:, 0] - t0) * TICK_US, (b[:, 1] - t0) * TICK_US          # us
            dur = en - st
            span = en.max()
            order = np.argsort(-dur)
            q = lambda a: [round(float(x), 2) for x in np.quantile(a, [0.5, 0.9, 0.99, 1.0])]
            alive = {f"{int(f * 100)}%": int(((st <= f * span) & (en > f * span)).sum()) for f in (0.1, 0.25, 0.5, 0.75, 0.9)}
            top = [dict(dur_us=round(float(dur[i]), 2), start_us=round(float(st[i]), 2), staged=int(b[i, 2]), visits=int(b[i, 3]),
                        us_per_staged=round(float(dur[i] / max(b[i, 2], 1)), 4)) for i in order[:6]]
            # per-SIMD / per-XCD load: sum of the durations of the waves that ran there, and when the last one ended
            per_simd = np.bincount(simd_key, weights=dur); per_simd = per_simd[per_simd > 0]
            xcd_end = [round(float(en[xcc == x].max()), 1) if (xcc == x).any() else 0.0 for x in range(8)]
            xcd_waves = [int((xcc == x).sum()) for x in range(8)]
            kern_us = stage["fwd.composite" if which == 0 else "bwd.composite"] * 1e3
            print(json.dumps(dict(config=cfg, dir=tag, kernel_us_hipevents=round(kern_us, 1), simds_used=int(per_simd.size), simd_busy_pct=q(per_simd), xcd_end_us=xcd_end, xcd_waves=xcd_waves, waves=int(len(b)), span_us=round(float(span), 2), dur_us_pct=q(dur),
                                  start_us_pct=q(st), staged_pct=q(b[:, 2]), visits_pct=q(b[:, 3]), waves_alive_at=alive,
                                  sum_wave_us=round(float(dur.sum()), 1), longest=top,
                                  counters={k: cnt[tag][k] for k in ("staged", "visits", "block_visits", "lanes_ok")})), flush=True)
            if args.dump:
                np.save(f"{args.dump}_{cfg}_{tag}.npy", np.concatenate([buf[ran], np.nonzero(ran)[0].astype(np.uint32)[:, None]], axis=1))


if __name__ == "__main__":
    main()
