#!/usr/bin/env python3
"""Per-kernel table of the PMC passes tools/pmc_step.sh collected over the real step: for every kernel instantiation with a
non-trivial share of the GPU time, launches, average duration, HBM-side traffic per launch (2 x FETCH_SIZE + WRITE_SIZE,
MI355X_MICROARCH.md § HBM: gfx950 tallies 128-B read requests at 64 B; both counters report KB), the share of the time the MFMA
pipes were busy (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)) and the effective clock (GRBM_GUI_ACTIVE, summed over the 8 XCDs,
/ 8 / duration).  Launches shorter than 0.3 ms read a high clock (the guide's caveat) and are listed without one.
    python3 tools/pmc_step_summary.py gpurun_out/<root> [out.json]"""
import collections
import csv
import glob
import json
import os
import re
import sys

COUNTERS = ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("lr::", "")
    return re.sub(r"\(.*$", "", name)


def read_pass(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    per = {}
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] != counter:
            continue
        k = int(row["Dispatch_Id"])
        v = per.get(k)
        if v is None:
            per[k] = [short(row["Kernel_Name"]), float(row["Counter_Value"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6]
        else:
            v[1] += float(row["Counter_Value"])          # one row per XCD / instance: summed
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for name, val, ms in per.values():
        a = agg[name]
        a[0] += 1; a[1] += val; a[2] += ms
    return agg


def main():
    root = sys.argv[1]
    data = {c: read_pass(os.path.join(root, c), c) for c in COUNTERS}
    names = set.intersection(*[set(d) for d in data.values()])
    total = sum(data["GRBM_GUI_ACTIVE"][n][2] for n in names)
    rows = []
    for n in names:
        cnt = data["GRBM_GUI_ACTIVE"][n][0]
        ms = sum(data[c][n][2] for c in COUNTERS) / sum(data[c][n][0] for c in COUNTERS)
        g = {c: data[c][n][1] / data[c][n][0] for c in COUNTERS}
        cycles = g["GRBM_GUI_ACTIVE"] / 8.0
        rows.append({"kernel": n, "launches_per_pass": cnt, "share_of_gpu_time": data["GRBM_GUI_ACTIVE"][n][2] / total, "avg_ms": ms,
                     "traffic_GB_per_launch": (2.0 * g["FETCH_SIZE"] + g["WRITE_SIZE"]) * 1024.0 / 1e9,
                     "fetch_GB": 2.0 * g["FETCH_SIZE"] * 1024.0 / 1e9, "write_GB": g["WRITE_SIZE"] * 1024.0 / 1e9,
                     "traffic_TBps": (2.0 * g["FETCH_SIZE"] + g["WRITE_SIZE"]) * 1024.0 / (ms * 1e-3) / 1e12,
                     "mfma_busy_frac": g["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cycles) if cycles else None,
                     "effective_clock_ghz": cycles / (ms * 1e-3) / 1e9 if ms >= 0.3 else None})
    rows.sort(key=lambda r: -r["share_of_gpu_time"])
    rows = [r for r in rows if r["share_of_gpu_time"] >= 0.002]
    print("| kernel | launches | share | ms/launch | traffic GB/launch (fetch + write) | TB/s | MFMA busy | clock GHz |")
    print("|---|---|---|---|---|---|---|---|")
    for r in rows:
        clk = f"{r['effective_clock_ghz']:.2f}" if r["effective_clock_ghz"] else "-"
        print(f"| `{r['kernel'][:90]}` | {r['launches_per_pass']} | {r['share_of_gpu_time']:.1%} | {r['avg_ms']:.3f} | {r['traffic_GB_per_launch']:.2f} "
              f"({r['fetch_GB']:.2f} + {r['write_GB']:.2f}) | {r['traffic_TBps']:.2f} | {r['mfma_busy_frac']:.1%} | {clk} |")
    if len(sys.argv) > 2:
        json.dump({"what": "rocprofv3 --pmc passes of `bench.py --steps 2 --warmup 1 --profile-run` (tools/pmc_step.sh), per kernel instantiation", "kernels": rows},
                  open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
