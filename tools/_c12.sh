export TD_PYTEST_SEL="tests/test_bench_shapes_gpu.py tests/test_distributed_gpu.py tests/test_ops_gpu.py"
bash tools/gpu_call.sh r3c12 sel
timeout 600 python bench.py --force-ddp --cpu-frames 0 --roofline-steps 0 > gpurun_out/r3c12_ddp1.json 2> gpurun_out/r3c12_ddp1.err
timeout 600 python bench.py --force-ddp --no-text-stream --cpu-frames 0 --roofline-steps 0 > gpurun_out/r3c12_ddp1_linear.json 2> gpurun_out/r3c12_ddp1_linear.err
timeout 600 python bench.py --cpu-frames 0 --roofline-steps 0 > gpurun_out/r3c12_default.json 2> gpurun_out/r3c12_default.err
TD_CONV_TAP_UNIFORM=1 python tools/conv_table.py 8 > gpurun_out/r3c12_table8.log 2>&1
