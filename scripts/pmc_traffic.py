#!/usr/bin/env python3
"""HBM traffic per stage and per render from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
(scripts/pmc_passes.sh).  Correction per MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled; WRITE_SIZE is
exact for 16-B streaming stores and float atomics.  (Gather-type reads are uncalibrated: the doubled
figure is an upper bound for them.)   usage: pmc_traffic.py <pmc dir> <tag> <out.json>"""
import csv, glob, json, os, sys, collections

d, tag, out = sys.argv[1], sys.argv[2], sys.argv[3]
STAGES = [("composite_bwd_kernel", "bwd.composite"), ("composite_fwd_kernel", "fwd.composite"),
          ("preprocess_fwd_kernel", "fwd.preprocess"), ("pergauss_bwd_kernel", "bwd.pergauss"),
          ("do_hist_kernel", "fwd.depth_order+scan"), ("do_bucket_scan_kernel", "fwd.depth_order+scan"), ("do_scatter_kernel", "fwd.depth_order+scan"),
          ("do_local_sort_kernel", "fwd.depth_order+scan"), ("tl_", "fwd.sort"),
          ("emit_keys_kernel", "fwd.emit_keys"), ("tile_ranges", "fwd.ranges"), ("scan", "fwd.depth_order+scan"), ("onesweep", "fwd.sort"), ("histogram", "fwd.sort"), ("radix_sort", "fwd.sort"), ("rocprim", "fwd.sort")]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
steps = 0
for f in sorted(glob.glob(os.path.join(d, f"{tag}_pass*_counter_collection.csv"))):
    rows = list(csv.DictReader(open(f)))
    if not rows or rows[0]["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_ATOMIC_sum"):
        continue
    cname = next(r["Counter_Name"] for r in rows if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"))
    n_fwd = sum(1 for r in rows if "composite_fwd_kernel" in r["Kernel_Name"] and r["Counter_Name"] == cname)
    n_bwd = sum(1 for r in rows if "composite_bwd_kernel" in r["Kernel_Name"] and r["Counter_Name"] == cname)
    for r in rows:
        if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        for pat, st in STAGES:
            if pat in r["Kernel_Name"]:
                tot[st][r["Counter_Name"]] += float(r["Counter_Value"]) / max(n_bwd if st.startswith("bwd.") else n_fwd, 1)
                break
res = {}
for st, c in tot.items():
    rd = 2.0 * c.get("FETCH_SIZE", 0.0) * 1024.0
    wr = c.get("WRITE_SIZE", 0.0) * 1024.0
    res[st] = int(rd + wr)
    res[st + ".read"] = int(rd); res[st + ".write"] = int(wr)
res["per_render_total"] = int(sum(v for k, v in res.items() if isinstance(v, int) and "." in k and k.count(".") == 1))
# VALU pipe utilisation of the two compositing kernels from the SQ / GRBM passes (scripts/pmc_summary.py's JSON):
#   busy = SQ_ACTIVE_INST_VALU [quad-cycles] x 4 / (1024 SIMDs x kernel cycles),  kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs
try:
    summ = json.load(open(os.path.join(d, f"{tag}_summary.json")))
    for k, v in summ.items():
        for pat, st in (("composite_bwd_kernel", "bwd.composite"), ("composite_fwd_kernel", "fwd.composite")):
            if pat in k and v.get("GRBM_GUI_ACTIVE") and v.get("SQ_ACTIVE_INST_VALU"):
                res[st + ".valu_busy"] = round(v["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * v["GRBM_GUI_ACTIVE"] / 8.0), 4)
except Exception:
    pass
# tie the figures to the kernels they were measured on: bench.py quotes them only while the native sources are unchanged
import hashlib, time
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_h = hashlib.sha256()
_csrc = os.path.join(_root, "gaussian_transformer_amd", "csrc")
for _n in sorted(os.listdir(_csrc)):
    if _n.endswith((".hip", ".h")):
        _h.update(_n.encode()); _h.update(open(os.path.join(_csrc, _n), "rb").read())
res["kernel_source_sha"] = _h.hexdigest()[:16]
res["captured"] = time.strftime("%Y-%m-%d %H:%M:%S")
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
