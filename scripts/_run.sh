mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x --timeout 1500 > gpurun_out/t3.log 2>&1; tail -5 gpurun_out/t3.log
for w in c2 cnr30 c5; do
  BVGPU_CTILE=0 python scripts/ab_time.py $w
  python scripts/ab_time.py $w
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab3.log
bash scripts/prof2.sh cnr_ct cnr30
bash scripts/prof2.sh c5_ct c5
