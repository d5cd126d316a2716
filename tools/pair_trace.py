#!/usr/bin/env python3
"""Reads a FEMFCT_PAIR_TRACE dump (phase timestamps of k_strip_jacobi_pair_walk, launch 1) and prints the mean phase
lengths per patch position and the timeline of two workgroups that share a CU (wg, wg + nwg/2).
usage: pair_trace.py <file> <walkers>"""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64)
print("waves per pattern (0-7 specialised, 8 generic), all launch-1 calls:", raw[-16:-7].tolist())
a = raw[:-16].reshape(-1, 16, 5).astype(np.int64)
nwg = int(sys.argv[2])
a = a[:nwg]
t0 = a[a > 0].min()
used = a[:, :, 0] > 0
print("patches per walker:", used.sum(axis=1).min(), "..", used.sum(axis=1).max())
for j in range(16):
    m = used[:, j]
    if not m.any():
        break
    seg = a[m, j, :]
    load, bar, sweep, store = (seg[:, 1] - seg[:, 0]).mean(), (seg[:, 2] - seg[:, 1]).mean(), (seg[:, 3] - seg[:, 2]).mean(), (seg[:, 4] - seg[:, 3]).mean()
    print(f"patch {j}: start {np.mean(seg[:, 0] - t0) / 100:7.2f} us | load {load / 100:6.2f} barrier {bar / 100:5.2f} sweeps {sweep / 100:6.2f} store-issue {store / 100:5.2f} us  (n={m.sum()})")
print("total:", (a[used].max() - t0) / 100, "us")
for wg in (0, nwg // 2, 5, 5 + nwg // 2):
    row = a[wg]
    print(f"wg {wg}: " + " | ".join(f"{(r[0] - t0) / 100:.1f} L {(r[1] - t0) / 100:.1f} S {(r[2] - t0) / 100:.1f}-{(r[3] - t0) / 100:.1f}" for r in row if r[0] > 0))
