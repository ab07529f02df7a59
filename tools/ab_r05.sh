#!/bin/bash
# A/B on ONE box: this tree against the round-5 tree (105ce1b, unpacked under _ab_r05/ by tools/make_ab_r05.sh with bench.load_scene patched in), both timing
# the SAME frozen scene (configs/synthetic/bench_scene_v1.pt), runs taking turns.   bash tools/ab_r05.sh [rounds] [outdir]
R=${1:-3}; O=${2:-gpurun_out/ab_r05}; mkdir -p $O
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-config2 --no-mc --no-hbm-kernels --no-alt-mode"
for i in $(seq 1 $R); do
  (cd _ab_r05 && python bench.py $Q > ../$O/r05_$i.json 2> ../$O/r05_$i.log)
  python bench.py $Q > $O/head_$i.json 2> $O/head_$i.log
done
python - $O <<'PY'
import json, glob, sys
for tag in ("r05", "head"):
    for f in sorted(glob.glob(sys.argv[1] + "/%s_*.json" % tag)):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
            print("%-5s value %.4f it/s  %.3f ms/step  matrix %.5f TFLOP/step  plain %.2f ms  remesh_extra %.2f ms  converged/iter %s" % (
                tag, d["value"], d["ms_per_step"], d.get("matrix_tflop_per_step") or -1, d["remesh"]["plain_step_ms"],
                d["remesh"]["remesh_extra_ms"], d.get("rays_converged_per_iter")))
        except Exception as e:
            print(tag, f, "failed:", e)
PY
