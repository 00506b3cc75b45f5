cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
timeout 900 python tools/outlier_probe.py > gpurun_out/r3c/outlier_probe.log 2>&1
cat gpurun_out/r3c/outlier_probe.log | grep -v amdgpu.ids | tail -20
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_lora.py -q -x 2>&1 | tail -3
