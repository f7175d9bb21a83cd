#!/bin/bash
# GPU box: SQ counters of scripts/ab_time.py for one workload (two passes).  usage: scripts/pmc2.sh <tag> <workload> [env assignments...]
tag=$1; wl=$2; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc_$tag; mkdir -p /tmp/pmc_$tag
pass() { local name=$1; shift; env BVGPU_OVERLAP=0 "${ENVS[@]}" rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$tag/$name -o $name --output-format csv -- python $R/scripts/ab_time.py $wl 2 > /tmp/pmc_$tag/$name.log 2>&1; }
ENVS=("$@")
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
pass sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS
python $R/scripts/pmc_summary.py /tmp/pmc_$tag > $R/gpurun_out/pmc_$tag.txt 2>&1
grep -A18 "${KERNEL:-k_parse}" $R/gpurun_out/pmc_$tag.txt | head -${LINES_SHOWN:-60}
