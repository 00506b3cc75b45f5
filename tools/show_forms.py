#!/usr/bin/env python3
"""Tabulate gpurun_out/locked_forms.jsonl (written by the full-size golden tests, tests/conftest.py record_locked_form): which operand
form `.to('cuda')` locked per (golden, mode), its probe distances and the error against the reference.
    python tools/show_forms.py [path]"""
import json, os, sys
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "locked_forms.jsonl")
rows = [json.loads(l) for l in open(path) if l.strip()]
print(f"{'golden':34s} {'backbone':8s} {'form':28s} {'err vs ref':>10s} {'probe s':>8s}  distances to the strict form (max over the probe rows)")
for r in rows:
    d = r.get("distance_to_strict") or {}
    print(f"{r['golden']:34s} {r['backbone']:8s} {r['form']:28s} {r['abs_err_vs_reference']:10.2e} {(r.get('probe_seconds') or 0):8.2f}  "
          + "  ".join(f"{k.replace('strict-vision', 'sv').replace('default+single-tail', 'tail')}={v:.1e}" for k, v in d.items()))
