#!/bin/bash
# Collects PMC counters for one scan command in separate passes (rocprofv3 --pmc with --kernel-trace only,
# as gpurun requires).  usage: scripts/pmc.sh <outdir> <python script + args...>
set -u
OUT=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/$OUT"
cd /tmp
pass() { # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d "$R/$OUT/$name" -o "$name" --output-format csv -- python "$R/$SCRIPT" $ARGS > "$R/$OUT/$name.log" 2>&1
}
SCRIPT=$1; shift; ARGS="$*"
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
pass sq2 SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
pass fetch FETCH_SIZE
pass write WRITE_SIZE TCC_HIT TCC_MISS
pass tlb TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_TOTAL_ACCESSES TCP_TCC_READ_REQ
python "$R/scripts/pmc_summary.py" "$R/$OUT" > "$R/$OUT/summary.txt" 2>&1
cat "$R/$OUT/summary.txt"
