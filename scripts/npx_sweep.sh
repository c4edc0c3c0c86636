#!/bin/bash
# compositing stage times of a config under blocks-per-wave settings (latency- vs throughput-bound regimes)
out=gpurun_out/npx_sweep.txt
: > $out
for c in "$@"; do
  for f in 1 2; do for b in 1 2; do
    timeout -k 10 200 python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --opt fwd_blocks_per_wave=$f --opt bwd_blocks_per_wave=$b 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['roofline']['stages']
print('$c fwd_npx=$f bwd_npx=$b', d['ms_per_step'], 'fwd.composite', s['fwd.composite']['ms'], 'bwd.composite', s['bwd.composite']['ms'])" >> $out || exit 1
  done; done
done
cat $out
