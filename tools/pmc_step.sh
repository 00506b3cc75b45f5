#!/bin/bash
# PMC passes over the REAL step (one counter per pass, kernel-trace only beside it): every kernel of the timed workload as the engine
# launches it -- the SwiGLU / residual-add / RoPE-qkv / operand-out instantiations of the GEMM, both attention kernels, the norms.
#   tools/pmc_step.sh <out_root under gpurun_out> [bench.py arguments]      then tools/pmc_step_summary.py
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-pmc_step}
shift
ARGS=${@:-}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  mkdir -p $OUT/$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 $R/bench.py --steps 2 --warmup 1 --profile-run --no-cpu-baseline $ARGS > $OUT/$c.log 2>&1
done
cd $R && python3 tools/pmc_step_summary.py $OUT
