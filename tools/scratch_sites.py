#!/usr/bin/env python3
"""Where a kernel's spill code sits: every scratch_load / scratch_store of one kernel in hipcc's ISA listing with the loop
depth of its basic block (LLVM annotates block labels with `Depth=`).  Usage:
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -S --cuda-device-only -I fem-fct-pdeco_amd/csrc \
        fem-fct-pdeco_amd/csrc/kernels_strip.hip -o /tmp/ks.s
  python3 tools/scratch_sites.py /tmp/ks.s k_strip_jacobi_pair_walkILi6ELi8E"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and re.match(r"^_Z\w+:", l))
end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i])
depth, counts, sites = 0, {}, []
for i in range(start, end):
    l = lines[i]
    if l.startswith(".LBB"):
        m = re.search(r"Depth=(\d+)", l)
        depth = int(m.group(1)) if m else 0
        # continuation comment lines ("Parent Loop", "Child Loop") do not change the block's own depth
    m = re.search(r"\b(scratch_(?:load|store)_\w+)", l)
    if m:
        counts[(depth, m.group(1).split("_")[1])] = counts.get((depth, m.group(1).split("_")[1]), 0) + 1
        sites.append((i - start, depth, l.strip().split(";")[0].strip()))
vg = next((l for l in lines[end:end + 400] if ".num_vgpr" in l), "")
sc = next((l for l in lines[end:end + 400] if "ScratchSize" in l or ".private_seg_size" in l or "private_segment_fixed_size" in l), "")
print(f"kernel {key}: {end - start} lines of ISA; {vg.strip()}; {sc.strip()}")
for (d, kind), c in sorted(counts.items()):
    print(f"  loop depth {d}: {c} scratch {kind}s")
deepest = max((d for d, _ in counts), default=0)
print(f"  deepest loop with spill code: depth {deepest}")
if "-v" in sys.argv:
    for s in sites:
        print("   ", s)
