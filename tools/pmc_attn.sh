#!/bin/bash
# PMC passes of the attention kernels (one counter per pass, kernel-trace only beside it) over tools/attn_bench.py, then a table.
#   tools/pmc_attn.sh <out_root under gpurun_out>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-pmc_attn}
cd /tmp && export TMPDIR=/tmp
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES; do
  mkdir -p $OUT/$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 $R/tools/attn_bench.py > $OUT/$c.log 2>&1
done
cd $R && python3 tools/pmc_attn_summary.py $OUT
