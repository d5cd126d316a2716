#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench.py command)
into profiles/traffic.json + a per-kernel summary CSV.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports half of the
bytes of a streaming read (MI355X_MICROARCH.md, HBM section); the factor was confirmed on this
library's 8-byte-per-lane streams (k_cheb: 2*FETCH = 83.9 B/row against 80 B/row compulsory reads,
WRITE = 8.0 B/row against 8).  Only dispatches with the roofline mesh's grid are counted.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <n_rows> <out_prefix>
"""
import collections
import csv
import json
import os
import re
import statistics
import sys

# kernel -> class, separately for the product path at the roofline size (fused multi-sweep kernels, operator derived
# in the kernels) and for the fusions-off pass of bench.py (one-sweep kernels); both run in the same bench.py process
PRODUCT = {"k_build_low_sb": "build_low", "k_strip4_jacobi": "jacobi", "k_strip4_jacobi_walk": "jacobi", "k_strip_jacobi_pair_walk": "jacobi", "k_strip4_cheb_mass_walk": "cheb",
           "k_strip4_cheb_mass_int": "cheb", "k_dudt_rhs_sb": "dudt_rhs", "k_strip4_cheb_mass": "cheb",
           "k_strip4_cheb": "cheb", "k_tile_flux_limit": "flux", "k_tile_jacobi": "jacobi", "k_tile_cheb": "cheb",
           "k_tile4_jacobi": "jacobi", "k_tile4_cheb": "cheb"}
# a class launch that consists of two kernels: the main one and its companion (boundary ring of the patch grid)
COMPANION = {"k_strip4_cheb_mass_int": "k_strip4_cheb_mass"}
ONE_SWEEP = {"k_build_low": "build_low", "k_jacobi": "jacobi", "k_dudt_rhs": "dudt_rhs", "k_cheb": "cheb", "k_flux": "flux",
             "k_limit": "limit", "k_ops_solidbody": "assemble"}


def load(path, counter, min_grid):
    """Per kernel: counter values of the roofline-mesh launches only (the 256-thread instantiation of
    the row kernels, or a grid of at least min_grid work-items for the tile kernels)."""
    agg = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"\bk_[a-z_0-9]+", r["Kernel_Name"])
            if not m:
                continue
            # (the walking kernels run one persistent workgroup per CU and exist only on large meshes)
            large = "<7, 256" in r["Kernel_Name"] or int(r["Grid_Size"]) >= min_grid or m.group(0).endswith("_walk")
            if large:
                agg[m.group(0)].append(float(r["Counter_Value"]))
            elif m.group(0) in COMPANION.values() and int(r["Grid_Size"]) >= min_grid // 8:
                agg[m.group(0) + "+ring"].append(float(r["Counter_Value"]))     # the boundary-ring launch of a split class
    return agg


def main():
    fpath, wpath, n, prefix = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    min_grid = n // 16         # work-items: only the large-mesh launches (row kernels: at most 2048 blocks x 256 threads;
                               # tile kernels: up to 4 nodes per thread); the PMC runs use --batched= so that no batch of
                               # small trajectories reaches this size
    fa = load(fpath, "FETCH_SIZE", min_grid)
    wa = load(wpath, "WRITE_SIZE", min_grid)
    rows, traffic, traffic1 = [], {}, {}
    for k in sorted(fa):
        f = statistics.mean(fa[k]) * 1024
        w = statistics.mean(wa.get(k, [0.0])) * 1024
        hbm = 2 * f + w
        rows.append(dict(kernel=k, launches=len(fa[k]), fetch_size_bytes_raw=f, write_size_bytes=w,
                         hbm_bytes_per_launch=hbm, hbm_bytes_per_row=hbm / n))
        if k in PRODUCT and not (PRODUCT[k] == "cheb" and k != "k_strip4_cheb_mass_int" and "k_strip4_cheb_mass_int" in fa):
            traffic[PRODUCT[k]] = hbm
            if k in COMPANION and COMPANION[k] + "+ring" in fa:
                kr = COMPANION[k] + "+ring"
                traffic[PRODUCT[k]] += (2 * statistics.mean(fa[kr]) + statistics.mean(wa.get(kr, [0.0]))) * 1024
        if k in ONE_SWEEP:
            traffic1[ONE_SWEEP[k]] = hbm
    with open(prefix + "_pmc_traffic.csv", "w", newline="") as f:
        wr = csv.DictWriter(f, fieldnames=list(rows[0]))
        wr.writeheader()
        wr.writerows(rows)
    # stamped with the identity of the kernel sources the measured library was built from: bench.py echoes these
    # bytes as roofline.traffic only when its own build carries the same stamp
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import source_sha16
    tj = os.path.join(os.path.dirname(prefix), "traffic.json")
    data = json.load(open(tj)) if os.path.exists(tj) else {}
    data = {k: v for k, v in data.items() if isinstance(v, dict) and "bytes_per_launch" in v}   # drop unstamped records
    data[f"n{n}"] = {"source_sha16": source_sha16(), "bytes_per_launch": traffic, "one_sweep_bytes_per_launch": traffic1,
                     "method": "(2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch, separate rocprofv3 --pmc passes"}
    json.dump(data, open(tj, "w"), indent=1, sort_keys=True)
    for r in rows:
        print(f"{r['kernel']:22s} launches {r['launches']:4d}  HBM {r['hbm_bytes_per_row']:7.1f} B/row/launch")


if __name__ == "__main__":
    main()
