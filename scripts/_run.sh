cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for w in c2 cnr30 c5; do timeout 300 python scripts/ab_time.py $w 10 2>&1 | grep "| scan" | tail -1 | cut -c1-260; done
python scripts/cnr_scan_time.py 2>&1 | tail -2; timeout 300 python scripts/chunk_time.py 2>&1 | tail -6
