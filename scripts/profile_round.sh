#!/bin/bash
# One-stop profiling for a round: bench line, rocprofv3 kernel stats, PMC passes + traffic.
# usage (on the GPU box, from the repo root): scripts/profile_round.sh <tag>
set -e
tag=${1:-rXX}
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python3 bench.py --steps 30 --warmup 3 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err || { tail -5 gpurun_out/$tag/bench.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/$tag/bench_under_rocprof.json 2> gpurun_out/$tag/rocprof.err || { tail -5 gpurun_out/$tag/rocprof.err; exit 1; }
scripts/pmc_passes.sh $tag > /dev/null
python3 scripts/pmc_summary.py gpurun_out/pmc $tag > gpurun_out/$tag/pmc_summary.txt
python3 scripts/pmc_traffic.py gpurun_out/pmc $tag gpurun_out/$tag/pmc_traffic.json > /dev/null
cp gpurun_out/pmc/${tag}_summary.json gpurun_out/$tag/pmc_summary.json
cat gpurun_out/$tag/bench.json
