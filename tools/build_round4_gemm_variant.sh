#!/bin/bash
# A/B library for profiles/r05_bench_variant_round4_gemm_kernels_*.json: round 4's gemm_conv.hip (commit 57ebf65) compiled against this
# tree's headers and linked with this tree's other objects -> tubedetr_amd/lib/libtubedetr_hip_r4.so (select with TD_HIP_LIB=<path>).
# Run `python -m tubedetr_amd.build` first.  (The old file still defines td_pw_chain, which nothing calls any more.)
set -e
cd "$(dirname "$0")/.."
L=tubedetr_amd/lib
git show 57ebf65:tubedetr_amd/csrc/gemm_conv.hip > tubedetr_amd/csrc/gemm_conv_r4tmp.hip
trap 'rm -f tubedetr_amd/csrc/gemm_conv_r4tmp.hip' EXIT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -c tubedetr_amd/csrc/gemm_conv_r4tmp.hip -o $L/gemm_conv_r4.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libtubedetr_hip_r4.so $L/api.o $L/gemm_conv_r4.o $L/prep.o $L/elementwise.o $L/attention.o $L/resnet_exec.o $L/optim.o $L/criterion.o $L/stem.o $L/bottleneck.o $L/cross_attn.o
echo $L/libtubedetr_hip_r4.so
