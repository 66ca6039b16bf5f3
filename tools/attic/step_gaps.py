#!/usr/bin/env python3
"""Where one config-1 step's time goes between its kernels: reads a rocprofv3 --kernel-trace CSV of `bench.py --no-...` and prints the
median duration of each kernel of the step and the median gap in front of it (end of the previous kernel -> its start), and the
gap from the end of the path kernel to the start of the next step's Gram kernel (D2H copy, sync, host, first launch).
    rocprofv3 --kernel-trace --output-format csv -d /tmp/t -o t -- python3 bench.py --steps 200 --no-cpu-baseline --no-c5 --no-host --no-two-callers --no-rccl-check
    python3 tools/step_gaps.py /tmp/t/*/t_kernel_trace.csv"""
import csv, sys, statistics as st
rows = []
for path in sys.argv[1:]:
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
names = ["gram_ring_kernel", "moments_reduce", "finalize_kernel", "path_rows_kernel"]
seq = [(s, e, next((k for k in names if k in n), None)) for s, e, n in rows]
seq = [x for x in seq if x[2]]
dur = {k: [] for k in names}; gap = {k: [] for k in names}
for i in range(1, len(seq)):
    s, e, k = seq[i]
    ps, pe, pk = seq[i - 1]
    if names.index(k) == (names.index(pk) + 1) % 4:
        dur[k].append(e - s); gap[k].append(s - pe)
tot = 0.0
for k in names:
    if dur[k]:
        d, g = st.median(dur[k]) / 1e3, st.median(gap[k]) / 1e3
        tot += d + g
        print(f"{k:22s} duration {d:8.2f} us   gap in front {g:7.2f} us   ({len(dur[k])} samples)")
print(f"sum of medians: {tot:.2f} us per step")
