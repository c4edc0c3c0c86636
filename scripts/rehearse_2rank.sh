#!/bin/bash
# Two-rank rehearsal of bench.py's N > 1 path on ONE GPU (gloo backend: RCCL refuses two ranks on one device).
# Functional check of the exchange modes only -- the numbers mean nothing (both ranks share the GPU, gloo stages through the host).
set -u
cfg=${1:-cfg2_table_300k_800}
port=29511
for mode in "GSR_X=default" "GSR_ALLREDUCE=direct" "GSR_SPARSE=1 GSR_ALLREDUCE=direct" "GSR_SPARSE=1" "GSR_EXCHANGE=overlap"; do
  port=$((port+1))
  echo "== $mode"
  env $mode GSR_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline --config $cfg 2>gpurun_out/rehearse_err.log | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print(d['value'], d['ms_per_step'], json.dumps(d['exchange']))" || { echo FAILED; tail -5 gpurun_out/rehearse_err.log; }
done
