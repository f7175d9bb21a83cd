cd /root/repo
python scripts/cnr_scan_time.py 2>&1 | tail -2
timeout 300 python scripts/chunk_time.py 2>&1 | tail -6
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
