mkdir -p gpurun_out
BVGPU_TILE=2 timeout 100 python scripts/dbg_ctile.py 20000 400000 2>&1 | grep -v amdgpu.ids | tail -3
BVGPU_TILE=2 timeout 100 python scripts/dbg_ctile.py 1500000 30000000 2>&1 | grep -v amdgpu.ids | tail -3
for w in c2 cnr30; do
  BVGPU_TILE=2 timeout 200 python scripts/ab_time.py $w
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab6.log
BVGPU_TILE=2 timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/t6.log 2>&1; tail -3 gpurun_out/t6.log
