mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q -k "seg_equals or one_block or root_finder" > gpurun_out/r03_grouped_tests.log 2>&1; tail -4 gpurun_out/r03_grouped_tests.log
RECMV_ROOT_TRACE=1 timeout 300 python bench.py --steps 8 --warmup 2 --settle-iters 240 --no-cpu-baseline --no-mc --no-hbm-kernels --no-alt-mode --no-config2 2>/dev/null | grep rootfind | tail -24 > gpurun_out/r03_rootfind_trace.txt; cat gpurun_out/r03_rootfind_trace.txt
