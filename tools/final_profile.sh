#!/bin/bash
# usage (on the GPU box, from the repo root): tools/final_profile.sh <tag>
# The round's committed measurements: rocprofv3 kernel stats of the default workload + the three PMC passes
# (FETCH_SIZE | WRITE_SIZE | MFMA-busy; never combined with each other or with other trace domains), aggregated per kernel.
tag=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- python $R/bench.py --no-graph --steps 3 --warmup 2 --cpu-frames 0 --roofline-steps 1 > $R/gpurun_out/${tag}_stats.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1); [ "$name" = "SQ_VALU_MFMA_BUSY_CYCLES" ] && name=MFMA
  timeout 900 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/${tag}_pmc_$name -- python $R/bench.py --no-graph --steps 1 --warmup 1 --cpu-frames 0 --roofline-steps 0 > $R/gpurun_out/${tag}_pmc_$name.log 2>&1
  python $R/tools/pmc_collect.py agg /tmp/${tag}_pmc_$name $R/gpurun_out/${tag}_pmc_${name}_per_kernel.csv >> $R/gpurun_out/${tag}_pmc_$name.log 2>&1
done
find $R/gpurun_out/${tag}_stats -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${tag}_kernel_stats.csv \;
echo done
