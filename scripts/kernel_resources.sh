#!/bin/bash
# VGPR / SGPR / LDS / scratch / occupancy of the kernels of one source (recompiles it with -Rpass-analysis=kernel-resource-usage).
# usage: kernel_resources.sh bv_kernels.hip [pattern] [extra hipcc flags]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${1:-bv_kernels.hip}; PAT=${2:-.}; shift; shift
cd $ROOT/webgraph_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Rpass-analysis=kernel-resource-usage "$@" -c $SRC -o /dev/null 2>&1 | awk '
  /remark: Function Name:/ {name=$5} /remark:     VGPRs:/ {v=$4} /remark:     AGPRs:/ {a=$4} /remark:     TotalSGPRs:/ {s=$4} /ScratchSize/ {p=$5} /Occupancy \[waves\/SIMD\]:/ {o=$5} /LDS Size/ {l=$6; printf "vgpr %-4s agpr %-3s sgpr %-4s scratch %-5s occ %-2s lds %-7s %s\n", v, a, s, p, o, l, name}' | c++filt | grep -E "$PAT" | cut -c1-170
