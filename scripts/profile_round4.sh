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
