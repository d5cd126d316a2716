#!/usr/bin/env python3
"""Per-dispatch durations of one kernel from a rocprofv3 --kernel-trace CSV (tuning helper).
usage: trace_kernel.py <dir with *kernel_trace.csv> <kernel-name substring>"""
import csv
import glob
import os
import sys

path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(path))
     if sys.argv[2] in r["Kernel_Name"]]
d.sort()
print(len(d), "dispatches; durations (us):", " ".join(f"{x[1] / 1e3:.0f}" for x in d))
