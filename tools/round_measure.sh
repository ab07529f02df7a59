# Round-end evidence on one MI355X box: PMC passes over bench.py (-> profiles/rNN_pmc_loop.json), the kernel trace of the bench
# command, the driver's bench command, the default command, op census, phase overlap, the GPU suite, the trajectory report, the
# run-to-run reproducibility counts, a two-rank job on the one GPU.      bash tools/round_measure.sh r06 <commit sha>
R=${1:-r06}; SHA=${2:-unknown}
REPO=$PWD
O=$REPO/gpurun_out/$R; mkdir -p $O
export TMPDIR=/tmp
QUIET="--no-cpu-baseline --no-mc --no-serial-pass --no-hbm-kernels --no-config2"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o run -- python $REPO/bench.py --steps 3 --warmup 1 \
      $QUIET --no-kernel-events > /tmp/pmc_$c.log 2>&1
done
cd $REPO
python tools/pmc_loop.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE --measured-at "commit $SHA, $(date -u +%Y-%m-%dT%H:%MZ), bench.py --steps 3 --warmup 1 (frozen scene v1)" > $O/${R}_pmc_loop.json
cp $O/${R}_pmc_loop.json profiles/${R}_pmc_loop.json      # bench.py reads the newest profiles/r*_pmc_loop.json
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $REPO/bench.py $QUIET > $O/${R}_bench_line_traced_command.json 2> /tmp/trace.err
cd $REPO
(echo "# rocprofv3 --kernel-trace --stats -- python bench.py $QUIET   (commit $SHA, frozen scene v1)"; python tools/prof_summary.py /tmp/prof 45) > $O/${R}_bench_kernel_trace.txt 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${R}_bench_line_driver_command.json 2> $O/${R}_bench_driver.err
tail -3 $O/${R}_bench_driver.err
( time timeout 900 python bench.py > $O/${R}_bench_line_default_command.json 2> $O/${R}_bench_default.err ) 2> $O/${R}_bench_default_walltime.txt
python tools/op_census.py --top 80 2>&1 | grep -v "Warn\|amdgpu\|run_backward" > $O/${R}_op_census.txt
python tools/phase_overlap.py 2>&1 | grep -v "Warn\|amdgpu\|run_backward" > $O/${R}_phase_overlap.txt
python tools/remesh_breakdown.py 2>&1 | grep -v "Warn\|amdgpu" > $O/${R}_remesh_breakdown.txt
timeout 1200 python -m pytest tests -m gpu -q > $O/${R}_gpu_suite_final.txt 2>&1
tail -3 $O/${R}_gpu_suite_final.txt
timeout 600 python -m pytest tests/test_gpu_composite.py -q -s -k trajectory 2>&1 | grep "on the GPU" > $O/${R}_trajectory_gpu.txt
cut -c1-400 $O/${R}_trajectory_gpu.txt
timeout 300 python tools/erratum/loop_repro_inproc.py 50 "f32 side-stream,f32 one-ray-stream,f32 serial" 2>/dev/null | cut -c1-400 > $O/${R}_loop_repro_50.txt
cat $O/${R}_loop_repro_50.txt
for i in 1 2 3 4 5 6; do timeout 120 python tools/determinism_probe.py 3 2>/dev/null | md5sum | cut -c1-8; done | sort | uniq -c > $O/${R}_loop_repro_processes.txt
cat $O/${R}_loop_repro_processes.txt
RECMV_SHARE_GPU0=1 timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 $QUIET > $O/${R}_bench_line_2ranks_one_gpu.json 2> $O/${R}_bench_2ranks.err
tail -2 $O/${R}_bench_2ranks.err
python -c "
import json
d=json.loads(open('$O/${R}_bench_line_driver_command.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'tflop/step', d['matrix_tflop_per_step'], 'frac', d['roofline']['frac'], 'kernel-only', d['roofline'].get('frac_kernel_only'), 'traffic', d['roofline']['traffic'], d['roofline']['traffic_source'])
print('remesh', d['remesh'])
print('config2', d.get('config2'))
print('mc', d.get('mc_only_roofline'))
for k in d.get('hbm_kernels', []): print(k['kernel'][:70], k['us'], k['frac'])
print('cpu', d.get('cpu_baseline', {}).get('value'))
"
