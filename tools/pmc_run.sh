#!/bin/bash
# PMC passes of the dominant GEMM (one counter per pass, kernel-trace only beside it), then the summary bench.py reads.
#   tools/pmc_run.sh <out_root under gpurun_out> "<forms>"      e.g. tools/pmc_run.sh pmc_r2 "mixed split single"
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-pmc_r2}
FORMS=${2:-mixed}
cd /tmp && export TMPDIR=/tmp
for form in $FORMS; do
  arg=$form; [ "$form" = single ] && arg=""
  for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
    mkdir -p $OUT/$form/$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$form/$c -- python3 $R/tools/gemm_one.py 4 $arg > $OUT/$form/$c.log 2>&1
  done
done
cd $R && python3 tools/pmc_summary.py $OUT $OUT/pmc_gemm_gate_up.json
