#!/bin/bash
# GPU box: the measurements committed under profiles/ for round 4.  Everything lands in gpurun_out/r4_*; every command under a timeout.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
export TMPDIR=/tmp
mkdir -p $O
cd $R
timeout 900 python bench.py > $O/r4_bench.json 2> $O/r4_bench.err; tail -c 600 $O/r4_bench.json; echo
timeout 600 python bench.py --workload C5 --no-extras > $O/r4_bench_c5.json 2>> $O/r4_bench.err
cd /tmp
for mode in overlapped serial; do
  rm -rf /tmp/prof_$mode
  if [ $mode = serial ]; then export BVGPU_OVERLAP=0; else unset BVGPU_OVERLAP; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode -o res -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-extras > /tmp/prof_$mode.log 2>&1
  python $R/scripts/rocprof_summary.py $(find /tmp/prof_$mode -name "*.db" | head -1) $O/r4_kernel_stats_$mode.txt
done
unset BVGPU_OVERLAP
# the segment pipeline (off by default): per-kernel times and counters of the same scan with it on
for cfg in "seg BVGPU_SEG=2" "seg_flat BVGPU_SEG=2 BVGPU_FLAT=1" "seg_own BVGPU_SEG=2 BVGPU_SEG_HANDOVER=0"; do
  set -- $cfg; name=$1; shift
  rm -rf /tmp/prof_s
  timeout 600 env BVGPU_OVERLAP=0 "$@" rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o res -- python $R/scripts/ab_time.py c2 5 > $O/r4_${name}_ab.log 2>&1
  python $R/scripts/rocprof_summary.py $(find /tmp/prof_s -name "*.db" | head -1) $O/r4_${name}_kernel_stats_c2.txt
  timeout 300 env "$@" python $R/scripts/ab_time.py c2 10 2>&1 | grep "| scan" | tail -1 >> $O/r4_seg_scan_times.txt
done
timeout 300 python $R/scripts/ab_time.py c2 10 2>&1 | grep "| scan" | tail -1 >> $O/r4_seg_scan_times.txt
mkdir -p $O/r4_pmc
for p in "sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "sq2 SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU" "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $p; name=$1; shift
  timeout 600 env BVGPU_SEG=2 BVGPU_OVERLAP=0 rocprofv3 --kernel-trace --pmc "$@" -d $O/r4_pmc/$name -o $name --output-format csv -- python $R/scripts/ab_time.py c2 3 > $O/r4_pmc/$name.log 2>&1
done
cd $R
python scripts/pmc_summary.py $O/r4_pmc > $O/r4_seg_pmc_c2.txt 2>&1
rm -rf $O/r4_pmc
for wl in c5 cnr30; do timeout 300 python scripts/ab_time.py $wl 10 2>&1 | grep "| scan" | tail -1 >> $O/r4_seg_scan_times.txt; done
cat $O/r4_seg_scan_times.txt | cut -c1-200
# the C5 shard: per-kernel times (serial), timeline of an overlapped scan, the group class's phases with and without the pre-walk
LINES_SHOWN=0 bash scripts/kstats.sh r4c5 c5 > /dev/null 2>&1; cp $O/kstats_r4c5.txt $O/r4_kernel_stats_serial_c5.txt
cd /tmp; rm -rf /tmp/tl
AB_NO_PROFILE=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/tl -o res -- python $R/scripts/ab_time.py c5 3 > /tmp/tl.log 2>&1
cd $R
python scripts/timeline.py $(find /tmp/tl -name "*.db" | head -1) > $O/r4_timeline_c5.txt 2>&1
for pw in 0 1; do echo "BVGPU_PREWALK=$pw" >> $O/r4_copybig_stats_c5.txt; BVGPU_PREWALK=$pw timeout 300 python scripts/copybig_stats.py c5 2>&1 | tail -12 >> $O/r4_copybig_stats_c5.txt; done
for pw in 0 2 1; do for wl in c5 c2 cnr30; do BVGPU_PREWALK=$pw timeout 300 python scripts/ab_time.py $wl 10 2>&1 | grep "| scan" | tail -1 >> $O/r4_prewalk_scan_times.txt; done; done
# hubs: rows of millions of successors, with and without the hand-over of their residuals (checked against the oracle)
for sizes in "8000000 4000000 2000000" "3000000" "1000000 1000000 500000 500000"; do for sg in 0 1; do HUB_CHECK=1 BVGPU_SEG=$sg timeout 600 python scripts/hub_time.py $sizes 2>&1 | tail -1 >> $O/r4_hubs.txt; done; done
cat $O/r4_hubs.txt $O/r4_prewalk_scan_times.txt | cut -c1-220
