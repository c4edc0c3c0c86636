#!/bin/bash
# One-stop profiling for a round: rocprofv3 kernel stats, PMC passes + traffic, then the bench line (which quotes the PMC
# traffic of the kernels it just ran: scripts/pmc_traffic.py stamps the JSON with the native-source hash bench.py checks).
# usage (on the GPU box, from the repo root): scripts/profile_round.sh <tag>     -> gpurun_out/<tag>/ ; copy into profiles/<tag>/
set -e
tag=${1:-rXX}
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/$tag/bench_under_rocprof.json 2> gpurun_out/$tag/rocprof.err || { tail -5 gpurun_out/$tag/rocprof.err; exit 1; }
scripts/pmc_passes.sh $tag > /dev/null
python3 scripts/pmc_summary.py gpurun_out/pmc $tag > gpurun_out/$tag/pmc_summary.txt
python3 scripts/pmc_traffic.py gpurun_out/pmc $tag gpurun_out/$tag/pmc_traffic.json > /dev/null
cp gpurun_out/pmc/${tag}_summary.json gpurun_out/$tag/pmc_summary.json
cp gpurun_out/$tag/pmc_traffic.json profiles/pmc_traffic.json          # the box's copy of the repo: the bench run below reads it
timeout -k 10 300 python3 bench.py --steps 30 --warmup 3 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err || { tail -5 gpurun_out/$tag/bench.err; exit 1; }
python3 scripts/lane_counters.py --npx 2 > gpurun_out/$tag/lane_counters.jsonl 2>/dev/null || true
cat gpurun_out/$tag/bench.json
