R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; export TMPDIR=/tmp
cd $R
python scripts/ab_time.py c2 3 > /dev/null 2>&1
for dm in 100000 1536 1024 512 256 128 64; do BVGPU_LW_DMAX=$dm python scripts/ab_time.py c2 5 2>&1 | tail -1 | cut -c40-60,120-300; done
