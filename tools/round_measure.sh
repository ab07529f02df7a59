# Round-end evidence on one MI355X box: PMC passes over bench.py (-> profiles/rNN_pmc_loop.json), the kernel trace of the bench
# command, the driver's bench command, the GPU suite.   bash tools/round_measure.sh r03 <commit sha>
R=${1:-r05}; SHA=${2:-unknown}
REPO=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o run -- python $REPO/bench.py --steps 3 --warmup 1 \
      --settle-iters 40 --no-cpu-baseline --no-mc --no-alt-mode --no-hbm-kernels --no-kernel-events --no-config2 > /tmp/pmc_$c.log 2>&1
done
cd $REPO
python tools/pmc_loop.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE --measured-at "commit $SHA, $(date -u +%Y-%m-%dT%H:%MZ), bench.py --steps 3 --warmup 1 --settle-iters 40" > gpurun_out/${R}_pmc_loop.json
cp gpurun_out/${R}_pmc_loop.json profiles/${R}_pmc_loop.json      # bench.py reads the newest profiles/r*_pmc_loop.json
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $REPO/bench.py --settle-iters 0 --no-alt-mode --no-cpu-baseline --no-mc --no-hbm-kernels --no-config2 > $REPO/gpurun_out/${R}_bench_line_traced_command.json 2> /tmp/trace.err
cd $REPO
(echo "# rocprofv3 --kernel-trace --stats -- python bench.py --settle-iters 0 --no-alt-mode --no-cpu-baseline --no-mc --no-hbm-kernels --no-config2   (commit $SHA)"; python tools/prof_summary.py /tmp/prof 45) > gpurun_out/${R}_bench_kernel_trace.txt 2>&1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_bench_line_driver_command.json 2> gpurun_out/${R}_bench_driver.err
tail -3 gpurun_out/${R}_bench_driver.err
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/${R}_gpu_suite_final.txt 2>&1
tail -3 gpurun_out/${R}_gpu_suite_final.txt
# row (g): the trajectory reports (14 / 35 iterations, the C2-sized pyramid) as the test prints them
timeout 600 python -m pytest tests/test_gpu_composite.py -q -s -k trajectory 2>&1 | grep "on the GPU:" > gpurun_out/${R}_trajectory_gpu.txt
cut -c1-600 gpurun_out/${R}_trajectory_gpu.txt
# run-to-run reproducibility: 50 repetitions of one iteration per cell, both matrix modes, side stream on / off
timeout 300 python tools/erratum/loop_repro_inproc.py 50 "f32 side-stream,f32 one-ray-stream,bf16x6 side-stream,bf16x6 one-ray-stream,f32 serial" 2>/dev/null | cut -c1-400 > gpurun_out/${R}_loop_repro_50.txt
cat gpurun_out/${R}_loop_repro_50.txt
# ... and across processes (f32, default switches): 6 fresh processes, 3 iterations each, one digest expected
for i in 1 2 3 4 5 6; do timeout 120 python tools/determinism_probe.py 3 2>/dev/null | md5sum | cut -c1-8; done | sort | uniq -c > gpurun_out/${R}_loop_repro_processes.txt
cat gpurun_out/${R}_loop_repro_processes.txt
python -c "
import json
d=json.loads(open('gpurun_out/${R}_bench_line_driver_command.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'whole', d['roofline']['whole_step']['frac_of_f32_mfma_peak'], 'traffic', d['roofline']['traffic'], d['roofline']['traffic_source'])
print('config2', d.get('config2'))
print('mc', d.get('mc_only_roofline'))
for k in d.get('hbm_kernels', []): print(k['kernel'][:70], k['us'], k['frac'])
print('cpu', d.get('cpu_baseline', {}).get('value'))
"
