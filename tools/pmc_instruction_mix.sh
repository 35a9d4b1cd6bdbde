#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_instruction_mix.sh <tag>
# Two PMC passes over one eager step of the default workload: instructions by class (VALU incl. MFMA, SALU, LDS, MFMA) and wave / wait / active
# cycles, aggregated per kernel -> gpurun_out/<tag>_pmc_instruction_mix_per_kernel.csv, <tag>_pmc_wave_cycles_per_kernel.csv.
# What it is for: VALU + SALU + LDS instructions per MFMA and SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES per kernel - a kernel at 4 - 5 other
# instructions per MFMA with two wavefronts per SIMD issues instructions, it does not wait for memory (the fused layer1 kernels, round 6).
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TD_ALLOW_RANDOM_TEXT_ENCODER=1
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  name=$([ $i = 0 ] && echo instruction_mix || echo wave_cycles); i=1
  timeout 900 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/${tag}_pmc_$name -- python $R/bench.py --no-graph --steps 1 --warmup 1 --cpu-frames 0 --roofline-steps 0 --dedupe-steps 0 > $O/${tag}_pmc_$name.log 2>&1
  python $R/tools/pmc_collect.py agg /tmp/${tag}_pmc_$name $O/${tag}_pmc_${name}_per_kernel.csv >> $O/${tag}_pmc_$name.log 2>&1
done
echo done
