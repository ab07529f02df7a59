"""Golden vectors for the two single-net root finders of utils/FindSurfacePs.py, from the REAL reference functions on CPU:

    python tests/golden/make_golden_rootfind_single.py        ->  tests/golden/rootfind_single.npz

  OptimizeGarmentSurfaceSinlge :210-272  one garment net, `offset_type` handed to the composite deformer; the reference calls it with
                                         dthreshold=1e-4, times=30 (OptimGarmentNetwork.py:2109, :2837, :3187, :3282)
  OptimizeSurfacePs            :145-207  the base-class loop's finder (OptimNetwork.py:523: dthreshold=5e-5, times=10; :268, :328:
                                         1e-4, 30).  It hands no `offset_type` to the deformer, and the reference's MLPTranslator
                                         indexes kwargs['offset_type'] (model/Deformer.py:177): with the garment deformer it can only
                                         raise KeyError there — the fixture runs it with the skinner alone as the deformer.

Inputs are the ones of rootfind.npz / translator.npz / lbs.npz (make_golden.py): same nets (rebuilt from the seed), same rays and start
points; only the outputs are new.
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import ref_loader  # noqa: E402

ref_loader.install()
import common_setup as cs  # noqa: E402


def main():
    torch.set_num_threads(8)
    N = ref_loader.ref_module("model.network")
    Dref = ref_loader.ref_module("model.Deformer")
    Fref = ref_loader.ref_module("utils.FindSurfacePs")
    g = {k: torch.from_numpy(v) for k, v in np.load(HERE / "rootfind.npz").items()}
    gt = {k: torch.from_numpy(v) for k, v in np.load(HERE / "translator.npz").items()}
    gl = {k: torch.from_numpy(v) for k, v in np.load(HERE / "lbs.npz").items()}
    ratio = {"sdfRatio": 0.8, "deformerRatio": 0.7, "renderRatio": 1.0}
    sdf = cs.build_sdf(N.getTmpSdf)
    tr = cs.build_translator(Dref.MLPTranslator)
    sk = cs.build_skinner(Dref.LBSkinner, Dref.batch_rodrigues)
    comp = Dref.CompositeDeformer([tr, sk])
    conds, poses, trans = gt["conds"], gl["poses"], gl["trans"]
    cam, rays, start, binds = g["cam_pos"], g["rays"], g["start"], g["binds"]
    out = {}
    # --- OptimizeGarmentSurfaceSinlge with the reference's call-site settings, and one step of it
    for tag, times in (("single", 30), ("single1", 1)):
        p, ok = Fref.OptimizeGarmentSurfaceSinlge(cam, rays, start.clone(), binds, sdf, ratio, comp, [conds, [poses, trans]],
                                                  dthreshold=1.e-4, athreshold=0.02, w1=3.05, w2=1., times=times,
                                                  offset_type="upper")
        out[tag + "_p"], out[tag + "_ok"] = p, ok
        print("OptimizeGarmentSurfaceSinlge times=%d: converged %d / %d" % (times, int(ok.sum()), ok.numel()))
    # --- OptimizeSurfacePs with the skinner alone: rays through the SKINNED start points (so that the finder can converge)
    with torch.no_grad():
        d0 = sk(start - 0.002 * cs.points(400, seed=15), [poses, trans], binds)
        rays_lbs = torch.nn.functional.normalize(d0 - cam.view(1, 3), dim=1)
    for tag, times in (("base", 10), ("base1", 1)):
        p, ok = Fref.OptimizeSurfacePs(cam, rays_lbs, start.clone(), binds, sdf, ratio, sk, [poses, trans], dthreshold=5.e-5,
                                       athreshold=0.02, w1=3.05, w2=1., times=times)
        out[tag + "_p"], out[tag + "_ok"] = p, ok
        print("OptimizeSurfacePs times=%d: converged %d / %d" % (times, int(ok.sum()), ok.numel()))
    try:
        Fref.OptimizeSurfacePs(cam, rays, start.clone(), binds, sdf, ratio, comp, [conds, [poses, trans]], times=1)
        raised = ""
    except KeyError as e:
        raised = "KeyError(%s)" % e
    print("OptimizeSurfacePs with the garment deformer:", raised or "ran")
    np.savez_compressed(HERE / "rootfind_single.npz", rays_lbs=rays_lbs.numpy(), composite_raises=np.array(raised),
                        **{k: v.detach().numpy() for k, v in out.items()})


if __name__ == "__main__":
    main()
