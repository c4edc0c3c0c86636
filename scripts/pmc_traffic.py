#!/usr/bin/env python3
"""HBM traffic per stage and per render from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
(scripts/pmc_passes.sh).  Correction per MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled; WRITE_SIZE is
exact for 16-B streaming stores and float atomics.  (Gather-type reads are uncalibrated: the doubled
figure is an upper bound for them.)   usage: pmc_traffic.py <pmc dir> <tag> <out.json>"""
import csv, glob, json, os, re, sys, collections

d, tag, out = sys.argv[1], sys.argv[2], sys.argv[3]
STAGES = [("composite_bwd_lpt_kernel", "bwd.composite"), ("composite_bwd_walk_kernel", "bwd.composite"), ("composite_bwd_pk_walk_kernel", "bwd.composite"), ("composite_fwd_walk_kernel", "fwd.composite"),
          ("pergauss_bwd_dense_kernel", "bwd.pergauss"), ("gather_visible_kernel", "bwd.second_stream"), ("fill_zero_kernel", "bwd.second_stream"),
          ("composite_bwd_kernel", "bwd.composite"), ("composite_bwd_pk_kernel", "bwd.composite"), ("zero_marked_rows_kernel", "bwd.clear+plan"), ("composite_fwd_kernel", "fwd.composite"),
          ("preprocess_fwd_kernel", "fwd.preprocess"), ("pergauss_bwd_kernel", "bwd.pergauss"),
          ("ss_sort_expand_kernel", "fwd.lists.order"), ("ss_count_kernel", "fwd.lists.bin"), ("ss_scan_kernel", "fwd.lists.bin"),
          ("ss_scatter_kernel", "fwd.lists.bin"),
          ("do_hist_kernel", "fwd.lists.bin"), ("do_bucket_scan_kernel", "fwd.lists.bin"), ("do_scatter_kernel", "fwd.lists.bin"),
          ("do_local_sort_kernel", "fwd.lists.bin"), ("tl_", "fwd.lists.order"),
          ("emit_keys_kernel", "fwd.lists.emit_keys"), ("tile_ranges", "fwd.lists.ranges"), ("scan", "fwd.lists.bin"), ("onesweep", "fwd.lists.order"), ("histogram", "fwd.lists.order"), ("radix_sort", "fwd.lists.order"), ("rocprim", "fwd.lists.order")]
tot = collections.defaultdict(lambda: collections.defaultdict(float))


def instrumented(name):
    """bench.py's single lane-counting render runs the <NPX, true, ...> compositing instantiations: not the timed kernels."""
    return "composite_" in name and re.search(r"kernel<\d+, [12]\b", name) is not None


import statistics
for f in sorted(glob.glob(os.path.join(d, f"{tag}_pass*_counter_collection.csv"))):
    rows = [r for r in csv.DictReader(open(f)) if not instrumented(r["Kernel_Name"])]
    if not rows or rows[0]["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_ATOMIC_sum"):
        continue
    cname = next(r["Counter_Name"] for r in rows if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"))
    is_fwd = lambda n: "composite_fwd_kernel" in n or "composite_fwd_walk_kernel" in n
    is_bwd = lambda n: any(k in n for k in ("composite_bwd_kernel", "composite_bwd_lpt_kernel", "composite_bwd_walk_kernel", "composite_bwd_pk_kernel", "composite_bwd_pk_walk_kernel"))
    n_fwd = sum(1 for r in rows if is_fwd(r["Kernel_Name"]) and r["Counter_Name"] == cname)
    n_bwd = sum(1 for r in rows if is_bwd(r["Kernel_Name"]) and r["Counter_Name"] == cname)
    per_kernel = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == cname:
            per_kernel[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    # median per dispatch of each kernel (bench.py's odd renders -- the reference-rule pair count, the instrumented one -- do not
    # skew it) x its launches per render
    for name, vals in per_kernel.items():
        for pat, st in STAGES:
            if pat in name:
                per_render = max(1, round(len(vals) / max(n_bwd if st.startswith("bwd.") else n_fwd, 1)))
                tot[st][cname] += statistics.median(vals) * per_render
                break
res = {}
for st, c in tot.items():
    rd = 2.0 * c.get("FETCH_SIZE", 0.0) * 1024.0
    wr = c.get("WRITE_SIZE", 0.0) * 1024.0
    res[st] = int(rd + wr)
    res[st + ".read"] = int(rd); res[st + ".write"] = int(wr)
res["per_render_total"] = int(sum(v for k, v in res.items() if isinstance(v, int) and "." in k and k.count(".") == 1))
# instruction counts per launch of the two compositing kernels (scripts/pmc_summary.py's JSON); bench.py prices the VALU count
# against the issue rate scripts/valu_rate.hip measures (one v_fma_f32 per 1.13 ns per SIMD, 1024 SIMDs) and the live kernel time
try:
    summ = json.load(open(os.path.join(d, f"{tag}_summary.json")))
    for k, v in summ.items():
        if instrumented(k):
            continue
        for pat, st in (("composite_bwd_lpt_kernel", "bwd.composite"), ("composite_bwd_walk_kernel", "bwd.composite"), ("composite_bwd_pk_walk_kernel", "bwd.composite"), ("composite_fwd_walk_kernel", "fwd.composite"),
                        ("composite_bwd_kernel", "bwd.composite"), ("composite_bwd_pk_kernel", "bwd.composite"), ("composite_fwd_kernel", "fwd.composite")):
            if pat in k and v.get("SQ_INSTS_VALU"):
                res[st + ".insts"] = {"valu": int(v["SQ_INSTS_VALU"]), "salu": int(v.get("SQ_INSTS_SALU", 0)), "lds": int(v.get("SQ_INSTS_LDS", 0)),
                                      "vmem_rd": int(v.get("SQ_INSTS_VMEM_RD", 0)), "vmem_wr": int(v.get("SQ_INSTS_VMEM_WR", 0)), "waves": int(v.get("SQ_WAVES", 0))}
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
res["workload"] = "cfg3_synth_1M_1080p"      # scripts/profile_round.sh runs the default bench.py under the counters
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
