#!/usr/bin/env python3
"""Folds the rocprofv3 --pmc passes of scripts/pmc_traffic.sh into per-kernel, per-launch averages.
HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KiB; the factor 2 is the gfx950 correction of
MI355X_MICROARCH.md "HBM": FETCH_SIZE tallies 128-byte requests at 64 bytes)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.match(r"(?:void )?(k_\w+)(<([\w, ]+)>)?", name)
    if not m:
        return None
    return m.group(1) + (("<" + m.group(3).replace(" ", "") + ">") if m.group(3) else "")


def fold(dirname):
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = short(row.get("Kernel_Name", ""))
                if not k:
                    continue
                c = row["Counter_Name"]
                acc[k][c] += float(row["Counter_Value"])
                calls[k][c] += 1
    return {k: {c: acc[k][c] / max(calls[k][c], 1) for c in acc[k]} for k in acc}, {k: max(calls[k].values()) for k in calls}


def main():
    out_dir = sys.argv[1]
    res = defaultdict(dict)
    launches = {}
    for sub in ("fetch", "write", "sq1", "sq2", "tcc"):
        per, n = fold(os.path.join(out_dir, sub))
        for k, d in per.items():
            res[k].update(d)
            launches[k] = n[k]
    for k, d in res.items():
        d["launches"] = launches.get(k, 0)
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_bytes_per_launch"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
        if "SQ_THREAD_CYCLES_VALU" in d and d.get("SQ_ACTIVE_INST_VALU"):
            d["active_lane_ratio"] = d["SQ_THREAD_CYCLES_VALU"] / (64.0 * d["SQ_ACTIVE_INST_VALU"])
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_* cycle counters tally quad-cycles over all SIMDs (MI355X_MICROARCH.md).
        # One wave64 VALU instruction holds its SIMD's issue for one quad-cycle, so
        #   VALU issue-slot utilisation = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)
        #   average waves per SIMD     = 4 * SQ_WAVE_CYCLES      / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)
        if d.get("GRBM_GUI_ACTIVE"):
            simd_cycles = 1024.0 * d["GRBM_GUI_ACTIVE"] / 8.0
            if "SQ_ACTIVE_INST_VALU" in d:
                d["valu_issue_util"] = 4.0 * d["SQ_ACTIVE_INST_VALU"] / simd_cycles
            if "SQ_WAVE_CYCLES" in d:
                d["occupancy_waves_per_simd"] = 4.0 * d["SQ_WAVE_CYCLES"] / simd_cycles
        if d.get("TCC_HIT_sum") is not None and (d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0)) > 0:
            d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    res["_meta"] = {"kernel_source_sha": bench.kernel_source_sha(), "config": os.environ.get("BENCH_CONFIG", "c2"),
                    "passes": "rocprofv3 --pmc, one counter set per run: FETCH_SIZE | WRITE_SIZE | SQ set 1 | SQ set 2 + GRBM | TCC/TCP",
                    "hbm_bytes_per_launch": "(2 x FETCH_SIZE + WRITE_SIZE) KiB, gfx950 correction of MI355X_MICROARCH.md"}
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
