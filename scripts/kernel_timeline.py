#!/usr/bin/env python
"""Timeline of the kernels of one bench step from a rocprofv3 --kernel-trace CSV: start offset, duration and queue of every
dispatch of the last complete step (a step = from one preprocess_fwd_kernel to the next).

    python scripts/kernel_timeline.py gpurun_out/kt/**/kt_kernel_trace.csv [--step -2]
"""
import csv
import glob
import sys


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/kt/**/*kernel_trace.csv"
    which = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else -2
    f = sorted(glob.glob(pat, recursive=True))[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "preprocess_fwd_kernel" in r["Kernel_Name"]]
    if len(starts) < 3:
        print("fewer than three steps in the trace"); return
    a, b = starts[which], starts[which + 1] if which + 1 < 0 or which + 1 < len(starts) else len(rows)
    t0 = int(rows[a]["Start_Timestamp"])
    prev_end = t0
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("gsr::", "").replace("void ", "")
        print(f'{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  (gap to previous end {(s - prev_end) / 1e3:7.1f})  q{r["Queue_Id"]}  {name[:90]}')
        prev_end = max(prev_end, e)
    print(f"step span {(prev_end - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
