#!/bin/bash
# A/B of two builds on one box: kernel stats of bench.py --quick under rocprofv3 for tools/dbg/lib_old.so and the current library
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in old new; do
  if [ $v = old ]; then export LLAVA_REWARD_HIP_LIB=$R/tools/dbg/lib_old.so; else unset LLAVA_REWARD_HIP_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$v -- python3 $R/bench.py --steps 2 --warmup 1 --quick $@ > $R/gpurun_out/ab_$v.log 2>&1
  echo "== $v: $(grep -o '"value": [0-9.]*' $R/gpurun_out/ab_$v.log | head -1)"
  f=$(ls $R/gpurun_out/ab_$v/*/*kernel_stats.csv | head -1)
  head -10 $f | cut -d, -f1-4,7 | sed 's/void lr:://; s/(lr::[A-Za-z]*Params)//' | cut -c1-150
done
