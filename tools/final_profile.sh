#!/bin/bash
# usage (on the GPU box, from the repo root): tools/final_profile.sh <tag>      e.g. r06
# The round's committed measurements, all from ONE call on ONE box:
#   1. rocprofv3 kernel stats of the default workload (eager launches, 6 steps)
#   2. the PMC passes (FETCH_SIZE | WRITE_SIZE | MFMA-busy | LDS bank conflicts; never combined with each other or with other trace
#      domains), aggregated per kernel, then profiles/<tag>_pmc_traffic.json / _pmc_mfma.json (what bench.py reads for roofline.traffic / mfma_util)
#   3. the default bench line (with its roofline and CPU legs) and one-flag variants of it; package power / clocks while it runs
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
export TD_ALLOW_RANDOM_TEXT_ENCODER=1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${tag}_stats -- python $R/bench.py --no-graph --steps 3 --warmup 2 --cpu-frames 0 --roofline-steps 1 --dedupe-steps 0 > $O/${tag}_bench_cfg3x16_under_rocprof.json 2> $O/${tag}_stats.err
find /tmp/${tag}_stats -name "*kernel_stats.csv" -exec cp {} $O/${tag}_bench_cfg3x16_kernel_stats.csv \;
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1); [ "$name" = "SQ_VALU_MFMA_BUSY_CYCLES" ] && name=MFMA; [ "$name" = "SQ_LDS_BANK_CONFLICT" ] && name=LDS
  timeout 900 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/${tag}_pmc_$name -- python $R/bench.py --no-graph --steps 1 --warmup 1 --cpu-frames 0 --roofline-steps 0 --dedupe-steps 0 > $O/${tag}_pmc_$name.log 2>&1
  python $R/tools/pmc_collect.py agg /tmp/${tag}_pmc_$name $O/${tag}_pmc_${name}_per_kernel.csv >> $O/${tag}_pmc_$name.log 2>&1
done
cd $R
python tools/pmc_traffic.py $O/${tag}_pmc_FETCH_SIZE_per_kernel.csv $O/${tag}_pmc_WRITE_SIZE_per_kernel.csv $O/${tag}_pmc_traffic.json > $O/${tag}_pmc_traffic.log 2>&1
python tools/pmc_collect.py mfma $O/${tag}_pmc_MFMA_per_kernel.csv $O/${tag}_pmc_mfma.json "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python bench.py --no-graph --steps 1 --warmup 1 --cpu-frames 0 --roofline-steps 0 --dedupe-steps 0" > $O/${tag}_pmc_mfma.log 2>&1
cp $O/${tag}_pmc_traffic.json $O/${tag}_pmc_mfma.json profiles/   # the bench line below reads them (this box's copy of the repo)
# the default line; rocm-smi sampled beside it (package power and clocks during the timed steps)
( for i in $(seq 1 60); do /opt/rocm/bin/rocm-smi --showpower --showclocks --json 2>/dev/null | head -c 700; echo; sleep 1; done ) > $O/${tag}_bench_cfg3x16_power_samples.log 2>&1 &
SMI=$!
timeout 900 python bench.py --steps 20 --warmup 5 > $O/${tag}_bench_cfg3x16.json 2> $O/${tag}_bench_cfg3x16.err
kill $SMI 2>/dev/null
V="--steps 10 --warmup 3 --cpu-frames 0 --roofline-steps 0 --dedupe-steps 0"
timeout 600 python bench.py $V > $O/${tag}_bench_variant_default.json 2>/dev/null
TD_CHAIN=0 timeout 600 python bench.py $V > $O/${tag}_bench_variant_chain_off.json 2>/dev/null     # layer3's conv3 -> next conv1 as two launches again
TD_WGRAD_WIDE4=1 timeout 600 python bench.py $V > $O/${tag}_bench_variant_wgrad_wide4.json 2>/dev/null  # weight gradients on four-wavefront 128 x 128 wave tiles
timeout 600 python bench.py $V > $O/${tag}_bench_variant_default_2.json 2>/dev/null
TD_CHAIN=0 timeout 600 python bench.py $V > $O/${tag}_bench_variant_chain_off_2.json 2>/dev/null
TD_CONV_BIG_PERSIST=0 timeout 600 python bench.py $V > $O/${tag}_bench_variant_big8_one_tile_per_workgroup.json 2>/dev/null
timeout 600 python bench.py $V --clips-per-gpu 1 > $O/${tag}_bench_variant_b1.json 2>/dev/null
timeout 600 python bench.py $V --clips-per-gpu 8 > $O/${tag}_bench_variant_b8.json 2>/dev/null
timeout 600 python bench.py $V --dedupe > $O/${tag}_bench_variant_b16_dedupe.json 2>/dev/null
timeout 600 python bench.py $V --no-graph > $O/${tag}_bench_variant_eager.json 2>/dev/null
timeout 600 python bench.py $V --force-ddp > $O/${tag}_bench_variant_ddp1.json 2>/dev/null
timeout 600 python bench.py $V --force-ddp --grad-collective rs_ag > $O/${tag}_bench_variant_ddp1_rs_ag.json 2>/dev/null
timeout 600 python bench.py --gpus 2 --oversubscribe --backend gloo --clips-per-gpu 4 --steps 3 --warmup 2 --cpu-frames 0 --roofline-steps 0 > $O/${tag}_bench_rehearsal_world2_gloo_one_gpu.json 2> $O/${tag}_bench_rehearsal_world2.err
timeout 600 python bench.py $V > $O/${tag}_bench_variant_default_again.json 2>/dev/null
for f in $O/${tag}_bench_*.json; do echo "$(basename $f): $(grep -o '"value": [0-9.]*' $f | head -1) $(grep -o '"ms_per_step": [0-9.]*' $f | head -1)"; done
echo done
