#!/bin/bash
# renders/s of every BASELINE configuration (bench.py --config ...), one JSON line each -> gpurun_out/all_configs.jsonl
# (config 3 is the metric; the others are recorded for context)
: > gpurun_out/all_configs.jsonl
for c in cfg1_plumbing_10k_256 cfg2_table_300k_800 cfg3_synth_1M_1080p cfg4_tiramisu_303k_1600x900 cfg5_stress_5M_4k; do
  timeout -k 10 300 python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline >> gpurun_out/all_configs.jsonl 2> gpurun_out/all_configs.err || { echo "failed: $c"; tail -3 gpurun_out/all_configs.err; exit 1; }
done
python3 - <<'PY'
import json
for l in open("gpurun_out/all_configs.jsonl"):
    b = json.loads(l); c = b["config"]
    print(f'{c["workload"]:32s} P={c["gaussians"]:8d} {c["width"]}x{c["height"]} pairs={c["num_rendered_pairs"]:9d}  {b["value"]:9.1f} renders/s  {b["ms_per_step"]:.4f} ms')
PY
