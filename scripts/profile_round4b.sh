#!/bin/bash
# GPU box: the measurements committed under profiles/ at the end of round 4 (after the straight-line one-lane loop).  Everything lands in gpurun_out/r4_*.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
export TMPDIR=/tmp
mkdir -p $O
cd $R
timeout 900 python bench.py > $O/r4_bench.json 2> $O/r4_bench.err; tail -c 400 $O/r4_bench.json; echo
timeout 600 python bench.py --workload C5 --no-extras > $O/r4_bench_c5.json 2>> $O/r4_bench.err
cd /tmp
for mode in overlapped serial; do
  rm -rf /tmp/prof_$mode
  if [ $mode = serial ]; then export BVGPU_OVERLAP=0; else unset BVGPU_OVERLAP; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode -o res -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-extras > /tmp/prof_$mode.log 2>&1
  python $R/scripts/rocprof_summary.py $(find /tmp/prof_$mode -name "*.db" | head -1) $O/r4_kernel_stats_$mode.txt
done
unset BVGPU_OVERLAP
cd $R
for wl in c5 cnr30; do LINES_SHOWN=0 bash scripts/kstats.sh r4$wl $wl > /dev/null 2>&1; cp $O/kstats_r4$wl.txt $O/r4_kernel_stats_serial_$wl.txt; done
for wl in c2 c5 cnr30; do
  cd /tmp; rm -rf /tmp/tl
  AB_NO_PROFILE=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/tl -o res -- python $R/scripts/ab_time.py $wl 3 > /tmp/tl.log 2>&1
  cd $R; python scripts/timeline.py $(find /tmp/tl -name "*.db" | head -1) > $O/r4_timeline_$wl.txt 2>&1
done
rm -f $O/r4_final_scan_times.txt
for wl in c2 c5 cnr30; do for r in 1 0; do BVGPU_LW_RES=$r timeout 300 python scripts/ab_time.py $wl 10 2>&1 | grep "| scan" | tail -1 >> $O/r4_final_scan_times.txt; done; done
cut -c1-230 $O/r4_final_scan_times.txt
