export TD_PYTEST_SEL="tests/test_ops_gpu.py::test_cross_attention_with_query_side_projections tests/test_ops_gpu.py::test_cross_attention_query_side_draws_the_same_dropout_mask_as_the_projected_path tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_distributed_gpu.py"
bash tools/gpu_call.sh r3c15 sel
timeout 600 python bench.py --cpu-frames 0 --roofline-steps 0 > gpurun_out/r3c15_default.json 2> gpurun_out/r3c15_default.err
TD_CROSS_Q1=0 timeout 600 python bench.py --cpu-frames 0 --roofline-steps 0 > gpurun_out/r3c15_projected.json 2> gpurun_out/r3c15_projected.err
timeout 600 python bench.py --cpu-frames 0 --roofline-steps 0 > gpurun_out/r3c15_default2.json 2> gpurun_out/r3c15_default2.err
TD_CROSS_Q1=0 timeout 600 python bench.py --cpu-frames 0 --roofline-steps 0 > gpurun_out/r3c15_projected2.json 2> gpurun_out/r3c15_projected2.err
python tools/conv_table.py 16 > gpurun_out/r3c15_table16.log 2>&1
