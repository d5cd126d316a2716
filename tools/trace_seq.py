"""Durations of the last large-grid kernel launches of a rocprofv3 kernel trace, in launch order (tuning helper)."""
import csv, sys, re
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    g = int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0)) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], g))
rows.sort()
# take the last 40 kernels whose grid is large
big=[(s,e,re.search(r"k_[a-z_0-9]+",n).group(0)) for s,e,n,g in rows if re.search(r"k_[a-z_0-9]+",n) and g>=1800*1024]
for s,e,n in big[-26:]:
    print(f"{n:24s} {(e-s)/1e3:8.1f} us")
