cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
timeout 2400 python -m pytest tests -m gpu -q -x -s 2>&1 > gpurun_out/r3g/pytest_full.log; tail -5 gpurun_out/r3g/pytest_full.log; grep -E "calibrate|outlier.*f16x2f8" gpurun_out/r3g/pytest_full.log | cut -c1-300 | head -20
