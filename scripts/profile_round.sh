#!/bin/bash
# GPU box: the measurements committed under profiles/ for one round.  usage: scripts/profile_round.sh <tag>
#   gpurun_out/<tag>_bench.json                  python bench.py (default flags)
#   gpurun_out/<tag>_kernel_stats_overlapped.txt rocprofv3 --kernel-trace --stats of the same command (streams as in production)
#   gpurun_out/<tag>_kernel_stats_serial.txt     the same with BVGPU_OVERLAP=0 (one kernel at a time: per-kernel durations)
#   gpurun_out/<tag>_timeline_overlapped.txt     kernel-by-kernel timeline of the last scan of the overlapped run
#   gpurun_out/<tag>_pmc/summary.txt             PMC counters, separate passes (scripts/pmc.sh)
tag=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 2500 gpurun_out/${tag}_bench.json
export TMPDIR=/tmp
cd /tmp
for mode in overlapped serial; do
  rm -rf /tmp/prof_$mode
  if [ $mode = serial ]; then export BVGPU_OVERLAP=0; else unset BVGPU_OVERLAP; fi
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode -o res -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-pmc > /tmp/prof_$mode.log 2>&1
  db=$(find /tmp/prof_$mode -name "*.db" | head -1)
  python $R/scripts/rocprof_summary.py $db $R/gpurun_out/${tag}_kernel_stats_$mode.txt
done
unset BVGPU_OVERLAP
# timeline of one production scan (bench.py ends with its serialised profiling passes, so use the tuning driver here)
rm -rf /tmp/prof_tl
TUNE_NO_PROFILE=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_tl -o res -- python $R/scripts/tune.py --reps 3 > /tmp/prof_tl.log 2>&1
python $R/scripts/timeline.py $(find /tmp/prof_tl -name "*.db" | head -1) $R/gpurun_out/${tag}_timeline_overlapped.txt
unset BVGPU_OVERLAP
head -12 $R/gpurun_out/${tag}_kernel_stats_serial.txt | cut -c1-140
cd $R
BVGPU_OVERLAP=0 scripts/pmc.sh gpurun_out/${tag}_pmc scripts/tune.py --reps 2 > /dev/null 2>&1
tail -5 gpurun_out/${tag}_pmc/summary.txt
# the checksum scan (hash folded inside the scan) against decode-then-fold: time, and bytes written (WRITE_SIZE, a pass of its own)
cd /tmp
python $R/scripts/checksum_time.py c2 > $R/gpurun_out/${tag}_checksum_time.txt 2>&1
python $R/scripts/checksum_time.py c5 >> $R/gpurun_out/${tag}_checksum_time.txt 2>&1
python $R/scripts/checksum_time.py cnr30 >> $R/gpurun_out/${tag}_checksum_time.txt 2>&1
for m in fold materialise; do
  rm -rf /tmp/prof_ck_$m
  CK_ONLY=$m BVGPU_OVERLAP=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_ck_$m -o ck --output-format csv -- python $R/scripts/checksum_time.py c2 3 > /tmp/prof_ck_$m.log 2>&1
  python $R/scripts/pmc_total.py /tmp/prof_ck_$m WRITE_SIZE >> $R/gpurun_out/${tag}_checksum_time.txt 2>&1
done
cat $R/gpurun_out/${tag}_checksum_time.txt
