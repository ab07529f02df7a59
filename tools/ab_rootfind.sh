mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q -k "seg_equals or one_block or root_finder" > gpurun_out/r03_grouped_tests.log 2>&1; tail -6 gpurun_out/r03_grouped_tests.log
for g in 0 1 0 1; do echo "RECMV_ROOT_GROUPED=$g"; RECMV_ROOT_GROUPED=$g timeout 120 python tools/loop_trace.py 10 2>&1 | tail -1; done
for g in 0 1; do echo "phases RECMV_ROOT_GROUPED=$g"; RECMV_TIMING=1 RECMV_ROOT_GROUPED=$g timeout 200 python bench.py --steps 10 --warmup 2 --settle-iters 40 --no-cpu-baseline --no-mc --no-hbm-kernels --no-alt-mode --no-config2 2>&1 >/dev/null | grep "phase ms"; done
