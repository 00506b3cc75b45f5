cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 > gpurun_out/r3b/pytest_full.log
grep -E "passed|failed|FAILED|outlier|e4m3|b2_ragged" gpurun_out/r3b/pytest_full.log | cut -c1-260 | tail -120
