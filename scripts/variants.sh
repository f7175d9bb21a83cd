#!/bin/bash
# Tuning builds of libbvgpu.so: scripts/variants.sh [-s source.hip] name "-DFOO=1 -DBAR=2" [name flags ...]  -> webgraph_amd/variants/libbvgpu_<name>.so
# Only `source` (default bv_seg.hip) is compiled with the flags; the other objects are the ones __graft_entry__.build() left in csrc/build.
# Select one at run time with BVGPU_LIB=<path>.
set -e
cd "$(dirname "$0")/../webgraph_amd/csrc"
src=bv_seg.hip
if [ "$1" = "-s" ]; then src=$2; shift 2; fi
stem=${src%.*}
mkdir -p ../variants
others=$(ls build/*.o | grep -v "build/$stem.o")
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $flags -c $src -o ../variants/${stem}_$name.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others ../variants/${stem}_$name.o -o ../variants/libbvgpu_$name.so && rm ../variants/${stem}_$name.o ) &
done
wait
ls -la ../variants
