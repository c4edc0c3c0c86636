#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (scripts/pmc_passes.sh): per kernel, mean counter value per dispatch."""
import csv, glob, sys, collections, json, os
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
tag = sys.argv[2] if len(sys.argv) > 2 else "run"
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, f"{tag}_pass*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0][-60:]
        res[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in res.items():
    if not any(s in k for s in ("gsr::", "rocprim")):
        continue
    out[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    out[k]["dispatches"] = max(len(v) for v in cs.values())
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    print(k)
    for c, x in sorted(v.items()):
        print(f"    {c:28s} {x:16.1f}")
json.dump(out, open(os.path.join(d, f"{tag}_summary.json"), "w"), indent=1)
