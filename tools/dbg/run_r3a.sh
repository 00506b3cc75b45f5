set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r3a/pytest.log
timeout 300 python tools/prec_map_probe.py 8 0 > gpurun_out/r3a/prec0.log 2>&1
timeout 300 python tools/prec_map_probe.py 8 2 > gpurun_out/r3a/prec2.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --quick --no-cpu-baseline > gpurun_out/r3a/bench_quick.json 2> gpurun_out/r3a/bench_quick.err
tail -5 gpurun_out/r3a/pytest.log; cat gpurun_out/r3a/prec0.log gpurun_out/r3a/prec2.log; cut -c1-400 gpurun_out/r3a/bench_quick.json
