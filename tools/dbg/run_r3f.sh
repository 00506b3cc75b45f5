cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
echo skip-tests
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3f/kstats -- python3 $R/bench.py --steps 3 --warmup 1 --quick --no-cpu-baseline > $R/gpurun_out/r3f/kstats.log 2>&1)
f=$(ls gpurun_out/r3f/kstats/*/*kernel_stats.csv | head -1); cp $f gpurun_out/r3f/r3_bench_kernel_stats_default.csv; head -8 $f | cut -c1-160
bash tools/pmc_run.sh r3f/pmc "mixed" 2>&1 | tail -3
bash tools/pmc_attn.sh r3f/pmc_attn 2>&1 | tail -12 | tee gpurun_out/r3f/r3_pmc_attention.md
timeout 1800 python bench.py --steps 8 --warmup 2 > gpurun_out/r3f/r3_bench.json 2> gpurun_out/r3f/bench.err; python tools/show_bench.py gpurun_out/r3f/r3_bench.json
rm -rf gpurun_out/r3f/kstats gpurun_out/r3f/pmc/mixed/*/ gpurun_out/r3f/pmc_attn/*/
