"""Row (g), second envelope: how far does the REFERENCE'S OWN 35-iteration loop move when every matrix product's result is
disturbed by a relative error of size eps — the size of a rounding difference, not of an algorithmic one?

tests/golden/trajectory_envelope.npz measures the spread under ONE kind of disturbance: torch-CPU sgemm with 1..8 threads, i.e. a
different partition of each product's reduction (last-bit changes of a few products' results).  The device's arithmetic differs from
the reference's in more places than that: every product is summed in the MFMA's order, the softplus / sine / rsqrt are the device's,
the sampler and the scatter-adds sum in another order.  All of these are rounding-sized (the kernel tests bound each against f64), but
they are MORE of them and somewhat larger than a re-partitioned sgemm, and the loop is chaotic: a larger seed
disturbance reaches a given distance earlier.  This script runs /root/reference's loop (make_golden_forward.main(trajectory=...), 4
sgemm threads like trajectory.npz) with torch.nn.functional.linear's result multiplied by (1 + eps * r), r uniform in [-1, 1) from a
seeded generator, for eps in EPS and seeds in SEEDS, and stores the canonical-mesh Chamfer distances of each run to trajectory.npz's
run per surface in tests/golden/trajectory_perturbed.npz.  tests/forward_case.py reports where the device's distances sit among them.

    python tests/golden/make_golden_perturbed.py [eps,eps,...] [seed,seed,...]          # ~6 minutes of host time per (eps, seed);
                                                                                        # rows are APPENDED to an existing file
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path[:0] = [str(HERE), str(HERE.parent), str(HERE.parent.parent)]
import forward_case as fc  # noqa: E402
import make_golden_forward as mg  # noqa: E402

EPS = (6e-8, 1e-6)        # one f32 ulp (2^-24) and ~16 ulp
SEEDS = (1, 2)


def perturbed_linear(eps, seed):
    base = torch.nn.functional.linear
    gen = torch.Generator().manual_seed(seed)

    def linear(x, w, b=None):
        y = base(x, w, b)
        r = torch.rand(y.shape, generator=gen, dtype=y.dtype) * 2.0 - 1.0
        return y * (1.0 + eps * r)
    return base, linear


def main():
    iters, period = fc.TRAJ_ITERS, 30
    eps_list = tuple(float(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else EPS
    seeds = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else SEEDS
    ref = np.load(HERE / "trajectory.npz")
    torch.set_num_threads(4)
    rows = []
    path = HERE / "trajectory_perturbed.npz"
    keys = ("eps", "seed", "losses", "canon_chamfer_body", "canon_chamfer_u", "canon_chamfer_b")
    out = {k: [] for k in keys}
    if path.is_file():
        old = np.load(path)
        out = {k: list(old[k]) for k in keys}
    for eps in eps_list:
        for seed in seeds:
            base, lin = perturbed_linear(eps, seed)
            torch.nn.functional.linear = lin
            try:
                print("=== reference loop, %d iterations, products disturbed by eps=%.0e (seed %d)" % (iters, eps, seed), flush=True)
                run = mg.main(trajectory=iters, remesh_period=period)
            finally:
                torch.nn.functional.linear = base
            d = {tag: fc.chamfer_vertices(run["canon_v_" + tag], torch.from_numpy(ref["canon_v_" + tag]))[0] for tag in ("body", "u", "b")}
            dl = np.abs(run["losses"].numpy() - ref["losses"]) / np.abs(ref["losses"])
            rows.append((eps, seed, d, dl))
            out["eps"].append(eps)
            out["seed"].append(seed)
            out["losses"].append(run["losses"].numpy())
            for tag in d:
                out["canon_chamfer_" + tag].append(d[tag])
            print("eps %.0e seed %d: canonical Chamfer to the undisturbed run  body %.3e  upper %.3e  bottom %.3e;  loss deviation "
                  "at iterations 5/15/25/35: %s" % (eps, seed, d["body"], d["u"], d["b"],
                                                    ["%.1e" % dl[min(i, len(dl) - 1)] for i in (4, 14, 24, 34)]), flush=True)
    np.savez_compressed(path, iters=np.asarray(iters), remesh_period=np.asarray(period),
                        **{k: np.asarray(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
