#!/usr/bin/env python3
"""Per-forward timeline from a rocprofv3 --kernel-trace CSV: the kernels of the LAST timed forward of bench.py in start order,
with the idle gaps on the device between them (usage: timeline.py <kernel_trace.csv> [marker-kernel-substring])."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = sys.argv[2] if len(sys.argv) > 2 else "text_encode_kernel"
starts = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
# the forward before the last text-encode (the last ones belong to the kernel-timing section of bench.py)
pick = starts[-3] if len(starts) >= 3 else starts[0]
end = starts[-2] if len(starts) >= 3 else len(rows)
t0 = int(rows[pick]["Start_Timestamp"])
last_end = t0
busy = []
for r in rows[pick:end]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {max(0, s - last_end) / 1e3:7.1f}  q{r.get('Queue_Id', '?'):>3}  {name}")
    last_end = max(last_end, e)
print(f"forward span {(last_end - t0) / 1e3:.1f} us")
