mkdir -p gpurun_out/r06i
python -m pytest tests/test_gpu_composite.py tests/test_gpu_loop.py -q -x > gpurun_out/r06i/pytest.txt 2>&1; tail -3 gpurun_out/r06i/pytest.txt
python tools/op_census.py --top 70 2>&1 | grep -v "Warn\|amdgpu\|run_backward" > gpurun_out/r06i/op_census.txt; head -3 gpurun_out/r06i/op_census.txt | cut -c1-160
