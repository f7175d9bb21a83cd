#!/bin/bash
# GPU box: the measurements committed under profiles/ for round 6 (the final code).  Everything lands in gpurun_out/r6p/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6p; mkdir -p $O
export TMPDIR=/tmp
cd $R
( time python bench.py --steps 20 --warmup 5 > $O/r6_bench.json 2> $O/r6_bench.err ) 2> $O/r6_bench.time; echo "bench rc=$? $(grep real $O/r6_bench.time)"
python bench.py --mode random > $O/r6_bench_random.json 2>/dev/null; echo "random rc=$?"
python bench.py --workload C5 --no-extras > $O/r6_bench_c5.json 2>/dev/null; echo "c5 rc=$?"
cd /tmp
for mode in overlapped serial; do
  rm -rf /tmp/prof_$mode
  if [ $mode = serial ]; then export BVGPU_OVERLAP=0; else unset BVGPU_OVERLAP; fi
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode -o res -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extras > /tmp/prof_$mode.log 2>&1
  python $R/scripts/rocprof_summary.py $(find /tmp/prof_$mode -name "*.db" | head -1) $O/r6_kernel_stats_$mode.txt
done
unset BVGPU_OVERLAP
head -8 $O/r6_kernel_stats_serial.txt | cut -c1-140
for wl in c2 c5 cnr30; do
  rm -rf /tmp/prof_tl
  rocprofv3 --kernel-trace -d /tmp/prof_tl -o res -- python $R/scripts/ab_time.py $wl 3 > /tmp/prof_tl.log 2>&1
  python $R/scripts/timeline.py $(find /tmp/prof_tl -name "*.db" | head -1) $O/r6_timeline_$wl.txt --back 3 > /dev/null
done
cd $R
bash scripts/pmc.sh gpurun_out/r6p/pmc scripts/ab_time.py c2 5 > /dev/null 2>&1
cp gpurun_out/r6p/pmc/summary.txt $O/r6_pmc_summary_c2.txt; rm -rf gpurun_out/r6p/pmc
python scripts/pmc_compare.py profiles/r5_pmc_summary_c2.txt $O/r6_pmc_summary_c2.txt > $O/r6_pmc_before_after_c2.txt; cut -c1-200 $O/r6_pmc_before_after_c2.txt | tail -25
for wl in c2 c5 cnr30; do python scripts/ab_time.py $wl 20 2>/dev/null | tail -1; done > $O/r6_final_scan_times.txt; cat $O/r6_final_scan_times.txt | cut -c1-300
