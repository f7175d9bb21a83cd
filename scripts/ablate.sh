#!/bin/bash
# What each part of a scan costs INSIDE the overlapped scan: tuning builds of libbvgpu.so that leave a part out (the records of an outdegree range are not decoded,
# the copy pass or one of its row classes is not launched -- wrong rows, right timing for everything else), timed by scripts/ab_time.py.
# usage (GPU box, after __graft_entry__.build()): scripts/ablate.sh [workload ...]     (default: c2 c5 cnr30)      -> one line per variant and workload
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
bash scripts/variants.sh -s bv_kernels.hip \
  ab_short "-DBV_EXP_DROP_LO=0 -DBV_EXP_DROP_HI=256" ab_lane "-DBV_EXP_DROP_LO=256 -DBV_EXP_DROP_HI=2048" ab_wave "-DBV_EXP_DROP_LO=2048 -DBV_EXP_DROP_HI=65536" \
  ab_giant "-DBV_EXP_DROP_LO=65536 -DBV_EXP_DROP_HI=0x7fffffff" ab_coop "-DBV_EXP_DROP_LO=2048 -DBV_EXP_DROP_HI=0x7fffffff" ab_parse "-DBV_EXP_DROP_LO=0 -DBV_EXP_DROP_HI=0x7fffffff" \
  ab_copy "-DBV_EXP_NOCOPY" ab_copylist "-DBV_EXP_NOCOPY_LIST" ab_copymid "-DBV_EXP_NOCOPY_MID" ab_copybig "-DBV_EXP_NOCOPY_BIG" \
  ab_all "-DBV_EXP_DROP_LO=0 -DBV_EXP_DROP_HI=0x7fffffff -DBV_EXP_NOCOPY" > /dev/null 2>&1
for w in ${@:-c2 c5 cnr30}; do
  python scripts/ab_time.py $w 2>&1 | grep "^$w" | sed 's/^/everything   /'
  for v in short lane wave giant coop parse copy copylist copymid copybig all; do
    BVGPU_LIB=$R/webgraph_amd/variants/libbvgpu_ab_$v.so python scripts/ab_time.py $w 2>&1 | grep "scan" | sed "s|BVGPU_LIB=[^ ]*||; s/^/without $v  /"
  done
done
