#!/bin/bash
# stage times of one config under combinations of native options: scripts/opt_matrix.sh <config> "a=1 b=2" "a=0" ...
cfg=$1; shift
out=gpurun_out/opt_matrix.txt
for combo in "$@"; do
  args=""
  for kv in $combo; do args="$args --opt $kv"; done
  timeout -k 10 200 python3 bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['roofline']['stages']
print('$cfg [$combo]', d['ms_per_step'], 'fwd.comp', s['fwd.composite']['ms'], 'bwd.zero', s['bwd.clear+plan']['ms'], 'bwd.comp', s['bwd.composite']['ms'], 'bwd.pg', s['bwd.pergauss']['ms'])" >> $out || exit 1
done
