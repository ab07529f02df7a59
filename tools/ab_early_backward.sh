# fixed scene (tools/loop_trace.py: 10 iterations from the initial state), two runs, then the default bench line
mkdir -p gpurun_out
for i in 1 2; do timeout 120 python tools/loop_trace.py 10 2>&1 | tail -1; done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mc --no-hbm-kernels --no-alt-mode --no-config2 > gpurun_out/r03_bench_c.json 2> gpurun_out/r03_bench_c.err; tail -2 gpurun_out/r03_bench_c.err
python -c "
import json; d=json.loads(open('gpurun_out/r03_bench_c.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step']['frac_of_f32_mfma_peak'], d['config']['mc_vertices'], d['rays_converged_fraction'], d['remesh'])"
