cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
for v in base st1 st2 base2; do
  if [ $v = base ] || [ $v = base2 ]; then unset LLAVA_REWARD_HIP_LIB; else export LLAVA_REWARD_HIP_LIB=$GRAFT_REPO_ROOT/llava-reward_amd/llava_reward_amd/libllava_reward_hip_$v.so; fi
  echo "=== $v"
  timeout 600 python tools/gemm_epi_probe.py 3 2>&1 | grep -v amdgpu.ids | cut -c1-120
  timeout 600 python bench.py --steps 4 --warmup 2 --quick --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', r['value'], r['ms_per_step'], r['parity_check']['abs_err'] if r.get('parity_check') else None)"
done 2>&1 | tee gpurun_out/r3e/ab.log
