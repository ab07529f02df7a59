set -x
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r2_gputests.log
RECMV_TIMING=1 timeout 600 python bench.py --steps 30 --warmup 1 --no-cpu-baseline --no-mc --no-alt-mode --no-hbm-kernels --settle-iters 120 > gpurun_out/r2_phases.json 2> gpurun_out/r2_phases.err
cd /tmp && export TMPDIR=/tmp
timeout 800 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 1 --no-cpu-baseline --no-mc --no-alt-mode --no-hbm-kernels --settle-iters 60 > /tmp/prof_bench.json 2> /tmp/prof_bench.err
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py /tmp/prof 70 > gpurun_out/r2_kernel_trace.txt 2>&1
