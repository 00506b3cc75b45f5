#!/usr/bin/env python3
"""Compare two rocprofv3 kernel_stats.csv files: python tools/stats_diff.py old.csv new.csv"""
import csv, re, sys
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        n = re.sub(r"\(.*", "", r["Name"].replace("void ", "").replace("lr::", ""))
        d[n] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["MaxNs"]) / 1e3)
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
ta, tb = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
print(f"{'kernel':60s} {'calls':>6s} {'old ms':>9s} {'new ms':>9s} {'delta':>7s} {'max us old/new':>18s}")
for n in sorted(set(a) | set(b), key=lambda k: -max(a.get(k, (0, 0, 0))[1], b.get(k, (0, 0, 0))[1]))[:16]:
    x, y = a.get(n, (0, 0, 0)), b.get(n, (0, 0, 0))
    print(f"{n[:60]:60s} {y[0]:6d} {x[1]:9.1f} {y[1]:9.1f} {100 * (y[1] - x[1]) / max(x[1], 1e-9):6.1f}% {x[2]:9.0f}/{y[2]:.0f}")
print(f"total {ta:.1f} -> {tb:.1f} ms ({100 * (tb - ta) / ta:+.2f} %)")
