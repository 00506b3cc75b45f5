#!/usr/bin/env python3
"""Print the essentials of a bench.py JSON line: python tools/show_bench.py gpurun_out/x.json"""
import json, sys
r = json.load(open(sys.argv[1]))
keep = ("value", "ms_per_step", "vs_headline", "parity", "preference_pairs_per_sec", "cores", "host_cpus", "seconds_per_row", "roofline_frac_whole_pass")
print("value", round(r["value"], 3), "ms_per_step", round(r["ms_per_step"], 1), "|", r["config"]["workload"])
for k, v in r.items():
    if isinstance(v, dict) and k != "config":
        d = {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in keep}
        pc = v.get("parity_check")
        if pc:
            d["abs_err"] = float("%.2e" % pc["abs_err"])
        print(" ", k, d)
rf = r["roofline"]
print("  roofline frac", round(rf["frac"], 4), "traffic", rf["traffic"], "whole_pass", rf.get("whole_pass"), (rf.get("dominant_kernel") or {}).get("pmc_note"))
if r.get("parity_check"):
    print("  parity_check", r["parity_check"]["abs_err"])
