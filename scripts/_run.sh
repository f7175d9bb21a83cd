cd /root/repo
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python scripts/cnr_scan_time.py 2>&1 | tail -2
for w in c2 c5 cnr30; do timeout 300 python scripts/ab_time.py $w 10 2>&1 | grep -v amdgpu.ids | tail -1; done
timeout 300 python scripts/deep_chains.py 2>&1 | tail -3
