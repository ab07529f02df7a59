"""Which phase of the iteration is not bit-reproducible in the bf16x6 matrix mode?  One process, one iteration's forward; then the
implicit differentiation is repeated on the SAME inputs, the gradients they add are
checksummed per tensor and restored.     RECMV_GEMM_MODE=1 python tools/bf16x6_phase_repro.py [reps]"""
import os
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    from recmv.hocon import ConfigFactory
    from recmv.loop import HotLoop
    dev = torch.device("cuda", 0)
    conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
    loop = HotLoop(conf, dev, stage="coarse", curves=True, **bench.HOTLOOP_KW)
    for it in range(2):
        loop.step(it)
    torch.cuda.synchronize()
    it = 2
    frame_ids = loop.frame_batch(it)
    ratio = {'sdfRatio': 1., 'deformerRatio': loop.opt_times / 2500. + 0.5, 'renderRatio': 1.}
    loop._allreduce = None
    loss = HotLoop.forward(loop, frame_ids, ratio)
    torch.cuda.synchronize()
    leaves = [("p%03d" % i, p) for i, p in enumerate(loop.shared_parameters())]
    leaves += [("TmpPs%d" % g, t) for g, t in enumerate(loop.TmpPs) if t is not None]

    def snapshot():
        return [None if p.grad is None else p.grad.detach().clone() for _, p in leaves]

    def restore(snap):
        for (_, p), g in zip(leaves, snap):
            p.grad = None if g is None else g.clone()

    def delta_sums(snap):
        out = []
        for (name, p), g in zip(leaves, snap):
            now = p.grad
            if now is None:
                out.append((name, "-"))
                continue
            d = now.detach().double() if g is None else (now.detach().double() - g.double())
            out.append((name, float(d.abs().sum()).hex()))
        return tuple(out)

    def study(label, fn):
        base = snapshot()
        seen = {}
        for r in range(reps):
            restore(base)
            torch.cuda.synchronize() if os.environ.get("PROBE_SYNC_BEFORE") == "1" else None
            fn()
            torch.cuda.synchronize()
            key = delta_sums(base)
            seen.setdefault(key, []).append(r)
        print("%s: %d distinct result(s) over %d repetitions: %s" % (label, len(seen), reps, [len(v) for v in seen.values()]), flush=True)
        keys = list(seen)
        for other in keys[1:]:
            diff = [a[0] for a, b in zip(keys[0], other) if a != b]
            print("   differs from the first in %d tensors: %s" % (len(diff), diff[:24]), flush=True)
        restore(base)
        return keys[0]

    loss.backward()                      # (not repeatable: the jet Functions free their workspaces) -> TmpPs gradients in place
    torch.cuda.synchronize()
    for joint in ("1", "0"):
        os.environ["RECMV_PROP_JOINT"] = joint
        study("propagateTmpPsGrad, RECMV_PROP_JOINT=%s" % joint, lambda: loop.propagateTmpPsGrad(frame_ids, ratio))


if __name__ == "__main__":
    main()
