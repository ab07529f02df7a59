# A/B of an environment switch on the bench line proper (settled scene): bash tools/ab_bench.sh VAR a b
VAR=$1; shift
for v in "$@" "$@"; do
  env $VAR=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mc --no-hbm-kernels --no-alt-mode --no-config2 --no-kernel-events 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['value'], d['ms_per_step'], d['config']['mc_vertices'], d['rays_converged_fraction'], d['remesh']['plain_step_ms'])"
done
