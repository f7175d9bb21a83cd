#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6aw; mkdir -p $O
export TMPDIR=/tmp; R=$PWD; cd /tmp
rm -rf /tmp/prof_tl; rocprofv3 --kernel-trace -d /tmp/prof_tl -o res -- python $R/scripts/ab_time.py c5 3 > /dev/null 2>&1
python $R/scripts/timeline.py $(find /tmp/prof_tl -name "*.db" | head -1) $R/$O/timeline_c5.txt --back 3 > /dev/null; grep -E "k_parse|k_copy|k_depth|k_scatter" $R/$O/timeline_c5.txt | cut -c1-100
cd $R; bash scripts/pmc.sh gpurun_out/r6aw/pmc scripts/ab_time.py c5 3 > /dev/null 2>&1; cp gpurun_out/r6aw/pmc/summary.txt $O/pmc_summary_c5.txt; rm -rf gpurun_out/r6aw/pmc
for k in "bv::k_copy_mid" "bv::k_copy_big" "bv::k_copy_prewalk<" "bv::k_copy_prewalk_long"; do awk -v K="$k" 'index($0,K)==1{p=1;print;next} /^bv::|^[a-z_A-Z]/{if(p)exit} p&&/SQ_WAVES|SQ_WAVE_CYCLES|SQ_INSTS_VALU|WAIT_ANY/{print}' $O/pmc_summary_c5.txt | cut -c1-100; done
