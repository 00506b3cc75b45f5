cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3j
timeout 3000 python -m pytest tests -m gpu -q -x -s > gpurun_out/r3j/pytest_full.log 2>&1; tail -3 gpurun_out/r3j/pytest_full.log; grep "w8a8 LLaVA" gpurun_out/r3j/pytest_full.log
timeout 1800 python bench.py --steps 8 --warmup 2 > gpurun_out/r3j/r3_bench.json 2> gpurun_out/r3j/bench.err; python tools/show_bench.py gpurun_out/r3j/r3_bench.json
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3j/kstats -- python3 $R/bench.py --steps 3 --warmup 1 --quick --no-cpu-baseline > $R/gpurun_out/r3j/kstats.log 2>&1)
f=$(ls gpurun_out/r3j/kstats/*/*kernel_stats.csv | head -1); cp $f gpurun_out/r3j/r3_bench_kernel_stats_default.csv; rm -rf gpurun_out/r3j/kstats
