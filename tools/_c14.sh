export TD_PYTEST_SEL="tests/test_ops_gpu.py::test_cross_attention_with_query_side_projections tests/test_ops_gpu.py::test_cross_attention_query_side_draws_the_same_dropout_mask_as_the_projected_path tests/test_model_gpu.py"
bash tools/gpu_call.sh r3c16 sel
timeout 600 python bench.py --cpu-frames 0 --roofline-steps 0 > gpurun_out/r3c16_default.json 2> gpurun_out/r3c16_default.err
TD_CROSS_Q1=0 timeout 600 python bench.py --cpu-frames 0 --roofline-steps 0 > gpurun_out/r3c16_projected.json 2> gpurun_out/r3c16_projected.err
timeout 600 python bench.py --cpu-frames 0 --roofline-steps 0 > gpurun_out/r3c16_default2.json 2> gpurun_out/r3c16_default2.err
python tools/conv_table.py 16 > gpurun_out/r3c16_table16.log 2>&1
