#!/bin/bash
# Recreate _ab_r05/ — the round-5 tree (commit 105ce1b) with bench.load_scene patched in — for tools/ab_r05.sh.  Run in the repository
# root of a clone that has the history; builds the round-5 library (hipcc, gfx950).     bash tools/make_ab_r05.sh
set -e
rm -rf _ab_r05 && mkdir -p _ab_r05
git archive 105ce1b rec-mv_amd include bench.py configs | tar -x -C _ab_r05
python - <<'PY'
src = open('bench.py').read()
a, b = src.index("SCENE_FILE = REPO /"), src.index("def export_state(loop, path, frame_ids, it):")
scene = src[a:b].replace('REPO / "configs" / "synthetic" / "bench_scene_v1.pt"', 'REPO.parent / "configs" / "synthetic" / "bench_scene_v1.pt"')
p = '_ab_r05/bench.py'
s = open(p).read()
s = s.replace("def export_state(loop, path, frame_ids, it):", scene + "def export_state(loop, path, frame_ids, it):", 1)
s = s.replace('ap.add_argument("--settle-iters", type=int, default=240,',
              'ap.add_argument("--scene", default=str(SCENE_FILE))\n    ap.add_argument("--settle-iters", type=int, default=0,')
old = "    it = 0\n    for _ in range(args.settle_iters):"
assert old in s
s = s.replace(old, '''    it = 0
    if args.scene != "none":
        it = load_scene(loop, args.scene, allreduce)
        sync()
        log("frozen scene loaded (r05 tree): MC vertices %s" % [int(v.shape[0]) for v in loop.garment_vs])
    torch.manual_seed(20261001 + rank)
    for _ in range(args.settle_iters):''')
s = s.replace('''        if gs:
            dom = max(gs, key=lambda k: gs[k]["seconds"])''', '''        line["mc_vertices_loop"] = [int(v.shape[0]) for v in loop.garment_vs]
        line["rays_converged_per_iter"] = round(converged / max(args.steps, 1), 1)
        if gs:
            line["matrix_tflop_per_step"] = round((sum(v["flops"] for v in gs.values()) + sum(v["gflop"] * 1e9 for v in (small_launches or {}).values())) / args.steps / 1e12, 5)
        if gs:
            dom = max(gs, key=lambda k: gs[k]["seconds"])''', 1)
open(p, 'w').write(s)
PY
(cd _ab_r05 && python rec-mv_amd/build.py)
echo "_ab_r05/ ready: bash tools/ab_r05.sh"
