# A/B of RECMV_SIDE_CUS (side streams confined to k of every 8 CUs): the bench scene is the same in every run (results do not depend on
# where waves run), so the step times compare directly.   bash tools/ab_side_cus.sh "0 3 4 5 6"
for k in ${1:-0 4}; do
  RECMV_SIDE_CUS=$k timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mc --no-hbm-kernels --no-alt-mode --no-config2 --no-kernel-events 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('RECMV_SIDE_CUS=$k  value %.3f it/s  %.2f ms  plain step %.2f ms  verts %s rays %s' % (d['value'], d['ms_per_step'], d['remesh']['plain_step_ms'], d['config']['mc_vertices'], d['config']['rays_per_iter']))
"
done
