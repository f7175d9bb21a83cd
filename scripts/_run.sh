mkdir -p gpurun_out
for w in c2 cnr30; do
  timeout 200 python scripts/ab_time.py $w
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab4.log
LINES_SHOWN=8 timeout 200 bash scripts/prof2.sh cnr_ct cnr30
LINES_SHOWN=8 timeout 200 bash scripts/prof2.sh c2_ct c2
