cd /root/repo
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 200 python scripts/host_path_time.py 2>&1 | tail -5
for v in "" "BVGPU_DBG=32"; do env $v timeout 300 python scripts/ab_time.py c5 5 2>&1 | grep -v amdgpu.ids | tail -1; done
timeout 300 python scripts/ab_time.py c2 10 2>&1 | tail -1
