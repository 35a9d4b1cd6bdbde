#!/bin/bash
# A/B builds of ONE source file with extra -D flags, linked against the stock objects of the other files:
#   tools/build_variant.sh <tag> <file.hip> "-DFOO=1 ..."  ->  tubedetr_amd/lib/libtubedetr_hip_<tag>.so   (select with TD_HIP_LIB=<path>)
set -e
cd "$(dirname "$0")/.."
tag=$1; src=$2; defs=$3
L=tubedetr_amd/lib
base=$(basename "$src" .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result $defs -c tubedetr_amd/csrc/$src -o $L/${base}_$tag.o
objs=""
for o in api gemm_conv prep elementwise attention resnet_exec optim criterion stem bottleneck cross_attn chain; do
  if [ "$o" == "$base" ]; then objs="$objs $L/${base}_$tag.o"; else objs="$objs $L/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libtubedetr_hip_$tag.so $objs
echo $L/libtubedetr_hip_$tag.so
