#!/bin/bash
# The profiles of a round, on the GPU box (one gpurun call):
#   1. kernel stats of the default-mode bench under rocprofv3 for the three backbones (`--quick`: only the timed workload's launches
#      plus the dominant-kernel probe's -- no golden sweep, no legs), 
#   2. PMC passes over the real Phi step (every kernel instantiation: traffic, MFMA busy, clock) -> <tag>_pmc_step.{md,json},
#   3. PMC passes of the dominant GEMM alone (-> <tag>_pmc_gemm_gate_up.json, which bench.py reads from profiles/) and of the attention kernels,
#   4. the full bench run (every leg; the compact driver line + gpurun_out/bench_legs.json) -> <tag>_bench.json / <tag>_bench_legs.json.
#   gpurun -- 'bash tools/profile_round.sh r4'        then copy gpurun_out/<tag>/<tag>_* into profiles/
set -u
TAG=${1:-rX}
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
for m in phi3v qwen llava; do
  B=32; [ $m = llava ] && B=64
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats_$m -- python3 $R/bench.py --model $m --batch $B --steps 3 --warmup 1 --profile-run --no-cpu-baseline > $R/$O/kstats_$m.log 2>&1)
  f=$(ls $O/kstats_$m/*/*kernel_stats.csv | head -1); cp $f $O/${TAG}_${m}_kernel_stats_default.csv; head -6 $f | cut -c1-150
  grep '^{' $O/kstats_$m.log | tail -1 > $O/${TAG}_${m}_kernel_stats_default.bench.json
done
bash tools/pmc_step.sh $TAG/pmc_step > $O/${TAG}_pmc_step.md 2>&1; python3 tools/pmc_step_summary.py $O/pmc_step $O/${TAG}_pmc_step.json > /dev/null; cat $O/${TAG}_pmc_step.md | cut -c1-220
bash tools/pmc_run.sh $TAG/pmc "mixed" 2>&1 | tail -3; cp $O/pmc/pmc_gemm_gate_up.json $O/${TAG}_pmc_gemm_gate_up.json
bash tools/pmc_attn.sh $TAG/pmc_attn 2>&1 | tail -12 | tee $O/${TAG}_pmc_attention.md
rm -rf $O/kstats_* $O/pmc/mixed/*/ $O/pmc_attn/*/ $O/pmc_step/*/
(cd $R && python3 bench.py --steps 8 --warmup 2 > $O/bench.log 2>&1); grep '^{' $O/bench.log | tail -1 > $O/${TAG}_bench.json; cp gpurun_out/bench_legs.json $O/${TAG}_bench_legs.json; cut -c1-600 $O/${TAG}_bench.json
