# A/B of the row-tile MLP passes on the bench line proper (settled scene), each configuration twice
run() {
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mc --no-hbm-kernels --no-alt-mode --no-config2 --no-kernel-events 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d['config']['mc_vertices'], d['rays_converged_fraction'], d['remesh']['plain_step_ms'])"
}
for rep in 1 2; do
  run RECMV_MLP_ROWS=0
  run RECMV_MLP_ROWS=1
  run RECMV_MLP_ROWS=1 RECMV_MLP_ROWS_MIN=1500 RECMV_MLP_ROWS_MAX=8192 RECMV_MLP_ROWS_RT=2
  run RECMV_MLP_ROWS=1 RECMV_MLP_ROWS_MIN=1024 RECMV_MLP_ROWS_MAX=4096 RECMV_MLP_ROWS_RT=1
done
