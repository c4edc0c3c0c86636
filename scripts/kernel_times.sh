#!/bin/bash
# rocprofv3 kernel averages of a short bench run (ns): scripts/kernel_times.sh [tag]    -> gpurun_out/kt_<tag>/
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/kt_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$tag -o s -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/kt_$tag.json 2> gpurun_out/kt_$tag.err || { tail -3 gpurun_out/kt_$tag.err; exit 1; }
python3 - <<PY
import csv, glob, json
f = glob.glob("gpurun_out/kt_$tag/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "gsr::" in r["Name"] and int(r["Calls"]) > 2: print(f'{r["Name"][:60]:60s} {r["Calls"]:>4s} {float(r["AverageNs"])/1e3:8.1f} us')
b = json.load(open("gpurun_out/kt_$tag.json")); print(b["value"], b["ms_per_step"])
PY
