#!/bin/bash
# rocprofv3 PMC passes over a short bench run (separate passes: gfx950 has 8 SQ + 4 TCC slots;
# FETCH_SIZE costs 3 TCC slots and WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Writes gpurun_out/pmc/<tag>_pass<k>_counter_collection.csv ; summarise with scripts/pmc_summary.py
set -e
tag=${1:-run}; shift || true
extra="$@"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
passes=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_WAVE_CYCLES"
 "FETCH_SIZE"
 "WRITE_SIZE TCC_EA0_ATOMIC_sum"
 "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
)
k=0
for p in "${passes[@]}"; do
  k=$((k+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $p --output-format csv -d gpurun_out/pmc -o ${tag}_pass$k -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $extra > gpurun_out/pmc/${tag}_pass$k.out 2> gpurun_out/pmc/${tag}_pass$k.err || { echo "pass $k failed"; tail -5 gpurun_out/pmc/${tag}_pass$k.err; }
done
ls gpurun_out/pmc | head -40
