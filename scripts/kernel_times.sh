#!/bin/bash
# per-kernel average durations of a bench run under rocprofv3: scripts/kernel_times.sh <config> [--opt a=b ...]
cfg=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/kt && mkdir -p gpurun_out/kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -o kt -- python3 bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/kt/bench.json 2> gpurun_out/kt/err.txt || { tail -5 gpurun_out/kt/err.txt; exit 1; }
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kt/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:22]:
    print(f'{float(r["AverageNs"])/1e3:9.1f} us x {int(r["Calls"]):5d}  {r["Name"][:110]}')
PY
