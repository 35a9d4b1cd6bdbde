#!/bin/bash
# PMC passes over tools/fused_l1_time.py (per-dispatch rows): where the fused stem / bottleneck kernels spend their time
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/l1pmc_$tag -- python $R/tools/fused_l1_time.py 1000 > $R/gpurun_out/l1pmc_$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
out=open(f"{R}/gpurun_out/l1pmc_summary.txt","w")
for d in sorted(glob.glob(f"{R}/gpurun_out/l1pmc_*/")):
    acc=collections.defaultdict(lambda: [0,0.0,0])
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n=r["Kernel_Name"]
            if not any(k in n for k in ("bottleneck","stem_pool","pw_resident","pw_chain","conv_gemm")): continue
            key=(n[:70], r["Counter_Name"])
            a=acc[key]; a[0]+=1; a[1]+=float(r["Counter_Value"])
            try: a[2]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
            except Exception: pass
    for (n,c),(k,s,t) in sorted(acc.items()):
        out.write(f"{n:72s} {c:24s} n={k:3d} avg={s/k:16.1f} avg_us={t/k/1e3:10.1f}\n")
out.close()
PY
