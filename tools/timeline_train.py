#!/usr/bin/env python3
"""Timeline of the LAST training step in a rocprofv3 --kernel-trace CSV of tools/train_bench.py: every kernel in start order with
its queue, and how long the device's main queue idles (usage: timeline_train.py <kernel_trace.csv> [min_us])."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
starts = [i for i, r in enumerate(rows) if "text_encode_kernel" in r["Kernel_Name"]]
pick, end = starts[-1], len(rows)
t0 = int(rows[pick]["Start_Timestamp"])
last_end, busy_any = t0, 0
for r in rows[pick:end]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
    gap = max(0, s - last_end)
    if (e - s) / 1e3 >= min_us or gap / 1e3 >= 20:
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {gap / 1e3:7.1f}  q{r.get('Queue_Id', '?'):>3}  {name}")
    busy_any += max(0, e - max(s, last_end))
    last_end = max(last_end, e)
print(f"step span {(last_end - t0) / 1e3:.1f} us; some kernel running {busy_any / 1e3:.1f} us")
