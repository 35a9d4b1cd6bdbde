#!/bin/bash
# usage: tools/gpu_call.sh <tag> [what...]   (runs on the GPU box from the repo root; everything lands in gpurun_out/<tag>_*)
tag=$1; shift
mkdir -p gpurun_out
for what in "$@"; do
  case $what in
    tests) timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_pytest.log ;;
    newtests) timeout 1200 python -m pytest tests/test_bench_shapes_gpu.py tests/test_fullsize_gpu.py tests/test_model_gpu.py tests/test_input_pipeline_gpu.py tests/test_optim_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/${tag}_newtests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_newtests.log ;;
    sel) timeout 1500 python -m pytest $TD_PYTEST_SEL -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/${tag}_sel.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_sel.log ;;
    bench) timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?" >> gpurun_out/${tag}_bench.err ;;
    benchfast) timeout 600 python bench.py --cpu-frames 0 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?" >> gpurun_out/${tag}_bench.err ;;
    stats) (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats -- python $GRAFT_REPO_ROOT/bench.py --no-graph --steps 3 --warmup 2 --cpu-frames 0 --roofline-steps 1 --dedupe-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats.log 2>&1) ;;
    *) echo "unknown step $what" ;;
  esac
done
ls gpurun_out | grep "^${tag}_" > /dev/null && echo done
