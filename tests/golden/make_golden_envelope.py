"""Row (g)'s envelope: how far apart do runs of the REFERENCE'S OWN loop end when only the summation order of its matrix products
changes?  tests/golden/trajectory.npz holds ONE pair (4 vs 1 sgemm threads); this script runs the reference's 35-iteration loop
(tests/golden/make_golden_forward.py main(trajectory=...), i.e. /root/reference's OptimGarmentNetwork.forward / propagateTmpPsGrad /
discretizeSDF through the stand-ins of ref_loader.py) K times — torch-CPU sgemm with 1, 2, 3, 4, 6 and 8 threads: six different
partitions of every product's reduction — and stores ALL pairwise canonical-mesh Chamfer distances per surface, the loss
trajectories and the final explicit meshes' distances in tests/golden/trajectory_envelope.npz.  The device test then holds the MI355X
run to the reference's measured spread (max pairwise distance) instead of a bound chosen by hand (tests/forward_case.py).

    python tests/golden/make_golden_envelope.py [iters=35] [remesh_period=30]        # ~K x several minutes of host time
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path[:0] = [str(HERE), str(HERE.parent), str(HERE.parent.parent)]
import forward_case as fc  # noqa: E402
import make_golden_forward as mg  # noqa: E402

THREADS = (1, 2, 3, 4, 6, 8)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else fc.TRAJ_ITERS
    period = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    name = sys.argv[3] if len(sys.argv) > 3 else "trajectory_envelope"
    runs = []
    for t in THREADS:
        torch.set_num_threads(t)
        print("=== reference loop, %d iterations, %d sgemm thread(s)" % (iters, t), flush=True)
        runs.append(mg.main(trajectory=iters, remesh_period=period))
    K = len(runs)
    out = {"threads": np.asarray(THREADS), "iters": np.asarray(iters), "remesh_period": np.asarray(period),
           "losses": np.stack([r["losses"].numpy() for r in runs])}
    for tag in ("body", "u", "b"):
        d = np.zeros((K, K))
        for i in range(K):
            for j in range(i + 1, K):
                d[i, j] = d[j, i] = fc.chamfer_vertices(runs[i]["canon_v_" + tag], runs[j]["canon_v_" + tag])[0]
        out["canon_chamfer_" + tag] = d
        out["canon_moved_" + tag] = np.asarray([float(r["canon_moved_" + tag][0]) for r in runs])
        out["canon_verts_" + tag] = np.asarray([int(r["canon_v_" + tag].shape[0]) for r in runs])
    for tag in ("u", "b"):
        d = np.zeros((K, K))
        for i in range(K):
            for j in range(i + 1, K):
                d[i, j] = d[j, i] = fc.chamfer_vertices(runs[i]["final_verts_" + tag], runs[j]["final_verts_" + tag])[0]
        out["explicit_chamfer_" + tag] = d
    # the run the device is compared with is trajectory.npz's (4 threads): its row of the matrices is what the test reads first
    out["reference_run"] = np.asarray(THREADS.index(4))
    np.savez_compressed(HERE / (name + ".npz"), **out)
    for tag in ("body", "u", "b"):
        d = out["canon_chamfer_" + tag]
        iu = np.triu_indices(K, 1)
        print("canonical %-4s pairwise Chamfer: min %.3e median %.3e max %.3e; against the 4-thread run: %s" % (
            tag, d[iu].min(), np.median(d[iu]), d[iu].max(), ["%.2e" % v for v in d[THREADS.index(4)]]))


if __name__ == "__main__":
    main()
