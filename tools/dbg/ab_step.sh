#!/bin/bash
# A/B of two builds on one box over the REAL step: kernel stats of `bench.py --quick` under rocprofv3 for tools/dbg/lib_old.so and the
# current library, alternating (new old new old) so that clock / thermal drift between runs shows up as the spread within a build.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in new1 old1 new2 old2; do
  case $v in old*) export LLAVA_REWARD_HIP_LIB=$R/tools/dbg/lib_old.so;; *) unset LLAVA_REWARD_HIP_LIB;; esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$v -- python3 $R/bench.py --steps 2 --warmup 1 --profile-run $@ > $R/gpurun_out/ab_$v.log 2>&1
done
python3 - <<'P'
import csv, glob, os, re
R = os.environ["GRAFT_REPO_ROOT"]
def load(v):
    d = {}
    for f in glob.glob(f"{R}/gpurun_out/ab_{v}/*/*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(.*", "", r["Name"].replace("void ", "").replace("lr::", ""))
            d[n] = d.get(n, 0.0) + float(r["TotalDurationNs"]) / 1e6
    return d
runs = {v: load(v) for v in ("new1", "old1", "new2", "old2")}
names = sorted(runs["old1"], key=lambda k: -runs["old1"][k])[:12]
print(f"{'kernel':52s} {'old1':>8s} {'old2':>8s} {'new1':>8s} {'new2':>8s}   new/old")
for n in names:
    o1, o2, n1, n2 = (runs[v].get(n, 0.0) for v in ("old1", "old2", "new1", "new2"))
    print(f"{n[:52]:52s} {o1:8.1f} {o2:8.1f} {n1:8.1f} {n2:8.1f}   {100 * ((n1 + n2) / max(o1 + o2, 1e-9) - 1):+5.1f} %")
t = {v: sum(runs[v].values()) for v in runs}
print(f"{'total':52s} {t['old1']:8.1f} {t['old2']:8.1f} {t['new1']:8.1f} {t['new2']:8.1f}   {100 * ((t['new1'] + t['new2']) / (t['old1'] + t['old2']) - 1):+5.1f} %")
P
