#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of the dominant GEMM into profiles/rN_pmc_gemm_gate_up.json (read by bench.py).

One pass per counter (MI355X_MICROARCH.md "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do not fit one pass), each written by

    rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -d gpurun_out/<root>/<form>/<COUNTER> -- python3 tools/gemm_one.py 4 [split|mixed]

for COUNTER in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE and form in single / split / mixed.  Then

    python3 tools/pmc_summary.py gpurun_out/<root> profiles/r2_pmc_gemm_gate_up.json

Per launch (first launch of a pass dropped: cold caches), as MI355X_MICROARCH.md § HBM prescribes:
    traffic      = 2 x FETCH_SIZE (gfx950 tallies 128-B read requests at 64 B) + WRITE_SIZE, both reported in KB
    clock        = GRBM_GUI_ACTIVE (sum over the 8 XCDs) / 8 / kernel duration
    mfma_busy    = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
The JSON records the hash of the kernel sources it was collected on; bench.py refuses the numbers for any other build."""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ("llava-reward_amd/csrc/gemm8.hip", "llava-reward_amd/csrc/common.h")      # keep in step with bench.py
COUNTERS = ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")


def source_sha16():
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()[:16]


def read_pass(d, counter):
    """-> (kernel name, [counter value per launch], [duration ms per launch]) for the gemm_bt8 dispatches of one pass."""
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    per = {}
    name = None
    for row in csv.DictReader(open(files[0])):
        if "gemm_bt8_kernel" not in row["Kernel_Name"] or row["Counter_Name"] != counter:
            continue
        name = row["Kernel_Name"]
        k = int(row["Dispatch_Id"])
        v, t0, t1 = per.get(k, (0.0, int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
        per[k] = (v + float(row["Counter_Value"]), t0, t1)          # one row per XCD / instance: summed
    ks = sorted(per)[1:]                                            # drop the first (cold) launch
    return name, [per[k][0] for k in ks], [(per[k][2] - per[k][1]) * 1e-6 for k in ks]


def main():
    root, out = sys.argv[1], sys.argv[2]
    forms = {}
    for form in ("single", "split", "mixed"):
        fd = os.path.join(root, form)
        if not os.path.isdir(fd):
            continue
        c, dur, name = {}, [], None
        for counter in COUNTERS:
            name, vals, ms = read_pass(os.path.join(fd, counter), counter)
            c[counter] = sum(vals) / len(vals)
            dur += ms
        avg_ms = sum(dur) / len(dur)
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0
        forms[form] = {"kernel_name": name, "launches_per_pass": len(dur) // len(COUNTERS), "avg_ms": avg_ms,
                       "fetch_size_kb": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"],
                       "traffic_bytes_per_launch": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0,
                       "mfma_busy_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"], "grbm_gui_active_sum_xcd": c["GRBM_GUI_ACTIVE"],
                       "mfma_busy_frac": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cycles),
                       "effective_clock_ghz": cycles / (avg_ms * 1e-3) / 1e9}
    json.dump({"what": "rocprofv3 --pmc passes of tools/gemm_one.py (decoder gate_up GEMM + SwiGLU, M=84544 N=16384 K=3072), per launch",
               "kernel_sources": list(KERNEL_SOURCES), "source_sha16": source_sha16(), "forms": forms}, open(out, "w"), indent=1)
    print(json.dumps(forms, indent=1))


if __name__ == "__main__":
    main()
