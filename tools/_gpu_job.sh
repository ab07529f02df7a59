timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2_gpu_tests_l.txt 2>&1
timeout 600 python tools/loop_trace.py 10 > gpurun_out/loop_proppre.txt 2>&1
timeout 600 python tools/loop_trace.py 10 >> gpurun_out/loop_proppre.txt 2>&1
timeout 300 python tools/determinism_probe.py 4 > gpurun_out/det_a.txt 2>gpurun_out/det_err.txt
RECMV_SERIAL=1 timeout 300 python tools/determinism_probe.py 4 > gpurun_out/det_c.txt 2>/dev/null
