export TD_PYTEST_SEL="tests/test_bench_shapes_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py"
bash tools/gpu_call.sh r3c13 sel
timeout 600 python bench.py --cpu-frames 0 --roofline-steps 0 > gpurun_out/r3c13_default.json 2> gpurun_out/r3c13_default.err
timeout 600 python bench.py --force-ddp --cpu-frames 0 --roofline-steps 0 > gpurun_out/r3c13_ddp1.json 2> gpurun_out/r3c13_ddp1.err
python tools/conv_table.py 8 > gpurun_out/r3c13_table8.log 2>&1
python tools/copy_sources.py 16 > gpurun_out/r3c13_copies16.log 2>&1
