#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of a bench.py run: the kernel sequence of ONE decoder layer of the last full-batch pass
(name, grid, duration), to see what each launch of a layer costs.   python tools/trace_layer.py <kernel_trace.csv> [layer_index]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(.*", "", n.replace("void ", "").replace("lr::", ""))
    return n[:70]
# find decoder attention launches (HD 96 causal) of the last big pass
att = [i for i, r in enumerate(rows) if "attn_kernel" in r["Kernel_Name"] and ", 96, true" in r["Kernel_Name"]]
big = [i for i in att if int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) > 2e6]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
i0, i1 = big[-32 + k], big[-32 + k + 1]
t_prev = None
tot = 0
for r in rows[i0:i1]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    gap = (int(r["Start_Timestamp"]) - t_prev) / 1e3 if t_prev else 0.0
    t_prev = int(r["End_Timestamp"])
    tot += d + max(gap, 0)
    gs, wg = r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))
    print(f"{short(r['Kernel_Name']):72s} grid {gs:>9s} wg {wg:>4s}  {d:9.1f} us  (gap {gap:6.1f})")
print(f"layer total {tot / 1e3:.3f} ms")
