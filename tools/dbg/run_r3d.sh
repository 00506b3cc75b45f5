cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r3d/pytest.log
tail -6 gpurun_out/r3d/pytest.log
timeout 1500 python bench.py --steps 5 --warmup 2 > gpurun_out/r3d/bench.json 2> gpurun_out/r3d/bench.err
tail -3 gpurun_out/r3d/bench.err
python tools/show_bench.py gpurun_out/r3d/bench.json 2>/dev/null | head -60 || cut -c1-1500 gpurun_out/r3d/bench.json
