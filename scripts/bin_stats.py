#!/usr/bin/env python
"""Super-tile bins of a config as supertile_sort.hip sees them: entries per bin and the largest depth sub-bucket under the kernel's two
maps (linear in depth / linear in the depth bits), recomputed on the host from the lists the HIP forward pass produced."""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--config", default="cfg4_tiramisu_303k_1600x900"); a = ap.parse_args()
    from gaussian_transformer_amd import synth
    from tests.helpers import oracle_scene
    from tests.test_gpu_parity import _stage_dump
    sc = synth.make_config(a.config)
    S = oracle_scene(sc)
    d = _stage_dump(S)
    W, H = S.W, S.H
    gx, gy = (W + 15) // 16, (H + 15) // 16
    SX, SY = (gx + 3) // 4, (gy + 3) // 4
    depth_bits = d["depth"].view(np.uint32)
    rows = []
    for sy in range(SY):
        for sx in range(SX):
            ids = []
            for ty in range(sy * 4, min(gy, sy * 4 + 4)):
                for tx in range(sx * 4, min(gx, sx * 4 + 4)):
                    r0, r1 = d["ranges"][ty * gx + tx]
                    ids.append(d["point_list"][r0:r1])
            ids = np.unique(np.concatenate(ids)) if ids else np.zeros(0, np.uint32)
            n = len(ids)
            if n == 0:
                continue
            kb = depth_bits[ids].astype(np.uint64)
            dep = d["depth"][ids].astype(np.float32)
            kmin, kmax = kb.min(), kb.max()
            lin = np.minimum(((dep - dep.min()) * (np.float32(512) / max(np.float32(dep.max() - dep.min()), np.float32(1e-30)))).astype(np.int64), 511)
            logm = ((kb - kmin) * 512 // (kmax - kmin + 1)).astype(np.int64)
            hl, hg = np.bincount(lin, minlength=512), np.bincount(logm, minlength=512)
            use = hg if hl.max() > 96 else hl
            rows.append((n, int(hl.max()), int(hg.max()), int((use.astype(np.int64) ** 2).sum()), float(dep.min()), float(dep.max())))
    r = np.array([x[:4] for x in rows])
    print(json.dumps(dict(config=a.config, bins=len(rows), n_max=int(r[:, 0].max()), n_sum=int(r[:, 0].sum()),
                          largest_subbucket_linear_max=int(r[:, 1].max()), largest_subbucket_log_max=int(r[:, 2].max()),
                          compares_total=int(r[:, 3].sum()), compares_max_bin=int(r[:, 3].max()),
                          worst_bins=sorted(rows, key=lambda x: -x[3])[:5])))


if __name__ == "__main__":
    main()
