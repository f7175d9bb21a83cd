#!/bin/bash
cd "$(dirname "$0")/.."
bash scripts/pmc.sh gpurun_out/r6x/pmc scripts/ab_time.py cnr30 5 > /dev/null 2>&1
cp gpurun_out/r6x/pmc/summary.txt gpurun_out/r6x/pmc_summary_cnr30.txt; rm -rf gpurun_out/r6x/pmc
grep -E "kernel|k_copy_list_w|k_copy_mid|k_copy_big|k_parse_tile" gpurun_out/r6x/pmc_summary_cnr30.txt | cut -c1-400
