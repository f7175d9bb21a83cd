#!/bin/bash
# Two SQ counter passes (rocprofv3 --pmc with --kernel-trace only) of scripts/ab_time.py for one library variant; prints the rows of the kernels matching a pattern.
# usage: scripts/pmc_ab.sh <outdir> <workload> <pattern> [lib.so]
set -u
OUT=$1; WL=$2; PAT=$3; LIB=${4:-}
export TMPDIR=/tmp AB_NO_PROFILE=1 BVGPU_OVERLAP=0
[ -n "$LIB" ] && export BVGPU_LIB=$LIB
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/$OUT"
cd /tmp
pass() { local name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d "$R/$OUT/$name" -o "$name" --output-format csv -- python "$R/scripts/ab_time.py" $WL 3 > "$R/$OUT/$name.log" 2>&1; }
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
pass sq2 SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS


python - "$R/$OUT" "$PAT" <<'PY'
import csv, glob, os, sys, re
from collections import defaultdict
root, pat = sys.argv[1], sys.argv[2]
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if not re.search(pat, name): continue
        agg[name.split("(")[0].replace("void ", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("    %-28s %14.1f  (%d dispatches)" % (c, sum(v) / len(v), len(v)))
PY
