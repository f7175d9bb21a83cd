mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x --timeout 1500 > gpurun_out/t2.log 2>&1; tail -5 gpurun_out/t2.log
for w in c2 cnr30 c5; do
  BVGPU_TILE=0 python scripts/ab_time.py $w
  python scripts/ab_time.py $w
  BVGPU_COOP_MIN=512 BVGPU_GIANT_MIN=32768 python scripts/ab_time.py $w
  BVGPU_COOP_MIN=1024 BVGPU_GIANT_MIN=32768 python scripts/ab_time.py $w
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab2.log
