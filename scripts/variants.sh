#!/bin/bash
# Tuning builds of libbvgpu.so: scripts/variants.sh name "-DFOO=1 -DBAR=2" [name flags ...]  -> webgraph_amd/variants/libbvgpu_<name>.so
# Select one at run time with BVGPU_LIB=<path>.
set -e
cd "$(dirname "$0")/../webgraph_amd/csrc"
mkdir -p ../variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result $flags bv_kernels.hip bv_offsets.hip bvgpu_api.cpp bvg_labels.cpp bv_host.cpp -o ../variants/libbvgpu_$name.so &
done
wait
ls -la ../variants
