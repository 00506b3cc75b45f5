#!/usr/bin/env python3
"""Table of the attention kernels' PMC passes (tools/pmc_attn.sh): per kernel instantiation and launch (first launch of a pass dropped),
duration, effective clock (GRBM_GUI_ACTIVE / 8 / duration), MFMA-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x
GRBM_GUI_ACTIVE / 8)), LDS index-unit activity and bank conflicts."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
data = collections.defaultdict(dict)
for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES"):
    files = glob.glob(os.path.join(root, c, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    per = collections.defaultdict(dict)
    for row in csv.DictReader(open(files[0])):
        if "attn_kernel" not in row["Kernel_Name"] or row["Counter_Name"] != c:
            continue
        k = int(row["Dispatch_Id"])
        v, t0, t1 = per[row["Kernel_Name"]].get(k, (0.0, int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
        per[row["Kernel_Name"]][k] = (v + float(row["Counter_Value"]), t0, t1)
    for name, d in per.items():
        ks = sorted(d)[1:]
        data[name][c] = (sum(d[k][0] for k in ks) / len(ks), sum((d[k][2] - d[k][1]) for k in ks) / len(ks) * 1e-6)
print("| kernel | ms / launch | clock GHz | MFMA pipes busy | LDS index unit active (of CU-cycles) | LDS bank-conflict cycles |")
print("|---|---|---|---|---|---|")
for name, d in data.items():
    if "GRBM_GUI_ACTIVE" not in d or "SQ_VALU_MFMA_BUSY_CYCLES" not in d:
        continue
    g, ms = d["GRBM_GUI_ACTIVE"]
    cyc = g / 8
    mf = d["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (1024 * cyc)
    lds = d.get("SQ_LDS_IDX_ACTIVE", (0, 0))[0] / (256 * cyc)
    bc = d.get("SQ_LDS_BANK_CONFLICT", (0, 0))[0]
    short = name.replace("void lr::", "").replace("(lr::AttnParams)", "")
    print(f"| `{short}` | {ms:.3f} | {cyc / (ms * 1e-3) / 1e9:.2f} | {100 * mf:.1f} % | {100 * lds:.1f} % | {bc:.3g} |")
