cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3h/k -- python3 $R/bench.py --model qwen --batch 32 --lora-rank 128 --steps 3 --warmup 1 --quick --no-cpu-baseline > $R/gpurun_out/r3h/log 2>&1)
f=$(ls gpurun_out/r3h/k/*/*kernel_stats.csv | head -1); cp $f gpurun_out/r3h/qwen_lora_kstats.csv; head -14 $f | cut -c1-170
f2=$(ls gpurun_out/r3h/k/*/*kernel_trace.csv | head -1)
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("$f2")))
# durations of the OUT_OP (epi 0) launches with small grids
d=collections.Counter(); t=collections.defaultdict(float)
for r in rows:
    if "gemm_bt8_kernel" in r["Kernel_Name"] and "0, 0, 2, 2" in r["Kernel_Name"]:
        g=int(r["Grid_Size_X"])//512 if "Grid_Size_X" in r else 0
        dur=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
        key=(g, round(dur,-1))
        d[g]+=1; t[g]+=dur
for g in sorted(d): print("workgroups",g,"launches",d[g],"avg us",round(t[g]/d[g],1),"total ms",round(t[g]/1e3,2))
PY
rm -rf gpurun_out/r3h/k
