#!/usr/bin/env python3
"""Per-kernel durations and inter-kernel gaps from a rocprofv3 --kernel-trace CSV (tuning helper):
python tools/trace_gaps.py <kernel_trace.csv> [min_calls]"""
import csv
import re
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
dur, gap_before, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
prev_end = None
for s, e, name in rows:
    m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", name)
    key = m.group(1) if m else name[:40]
    dur[key] += e - s
    cnt[key] += 1
    if prev_end is not None and 0 <= s - prev_end < 50_000:
        gap_before[key] += s - prev_end
    prev_end = e
minc = int(sys.argv[2]) if len(sys.argv) > 2 else 100
print(f"{'kernel':42s} {'calls':>7s} {'avg_us':>8s} {'gap_before_us':>14s}")
for k in sorted(cnt, key=lambda k: -dur[k]):
    if cnt[k] >= minc:
        print(f"{k:42s} {cnt[k]:7d} {dur[k] / cnt[k] / 1e3:8.2f} {gap_before[k] / cnt[k] / 1e3:14.2f}")
