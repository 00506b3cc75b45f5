#!/bin/bash
# The profiles of a round, on the GPU box (one gpurun call): kernel stats of the default-mode bench under rocprofv3, PMC passes of the
# dominant GEMM (-> pmc_gemm_gate_up.json, which bench.py reads from profiles/) and of the attention kernels, then the full bench line.
#   gpurun -- 'bash tools/profile_round.sh r3'        then copy gpurun_out/<tag>/{<tag>_*.{csv,md,json}, pmc/pmc_gemm_gate_up.json} into profiles/
set -u
TAG=${1:-rX}
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats -- python3 $R/bench.py --steps 3 --warmup 1 --quick --no-cpu-baseline > $R/$O/kstats.log 2>&1)
f=$(ls $O/kstats/*/*kernel_stats.csv | head -1); cp $f $O/${TAG}_bench_kernel_stats_default.csv; head -8 $f | cut -c1-160
bash tools/pmc_run.sh $TAG/pmc "mixed" 2>&1 | tail -3
bash tools/pmc_attn.sh $TAG/pmc_attn 2>&1 | tail -12 | tee $O/${TAG}_pmc_attention.md
timeout 1800 python bench.py --steps 8 --warmup 2 > $O/${TAG}_bench.json 2> $O/bench.err; python tools/show_bench.py $O/${TAG}_bench.json
rm -rf $O/kstats $O/pmc/mixed/*/ $O/pmc_attn/*/
