set -x
RECMV_SERIAL_RAYS=1 timeout 600 python bench.py --steps 30 --warmup 1 --no-cpu-baseline --no-mc --no-alt-mode --no-hbm-kernels > gpurun_out/r2_bench4_serial.json 2> gpurun_out/r2_bench4.err
timeout 600 python bench.py --steps 30 --warmup 1 --no-cpu-baseline --no-mc --no-alt-mode --no-hbm-kernels > gpurun_out/r2_bench4_overlap.json 2>> gpurun_out/r2_bench4.err
