#!/usr/bin/env python3
"""Where a kernel's register spills sit: per function of an assembly listing (hipcc -S), the scratch loads / stores
inside loops (a spill in a hot loop is a stall; in the prologue or the tail it is noise).  usage: spills.py file.s"""
import re, sys
fn, lab, out = None, "", {}
for n, line in enumerate(open(sys.argv[1]), 1):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        fn, lab = m.group(1), ""
    m = re.match(r"^(\.LBB\d+_\d+):\s*;?\s*(.*)", line)
    if m:
        lab = m.group(1) + " " + m.group(2).strip()
    if "scratch_" in line and fn:
        out.setdefault(fn, []).append((n, line.split()[0], lab))
for f, items in out.items():
    inloop = [i for i in items if "Loop" in i[2]]
    print(f"{f[:60]}: {len(items)} scratch ops, {len(inloop)} inside loops")
    for n, op, lab in inloop:
        print(f"   line {n}: {op:24s} {lab[:70]}")
