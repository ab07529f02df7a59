"""Run K iterations of the loop and print, per iteration, checksums of the loss terms and of the state — run twice and
diff the outputs to find the first iteration / term that depends on scheduling.

    python tools/determinism_probe.py [iters] > a.txt ; python tools/determinism_probe.py [iters] > b.txt ; diff a.txt b.txt
"""
import os
import struct
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import torch  # noqa: E402
from recmv.hocon import ConfigFactory  # noqa: E402
from recmv.loop import HotLoop  # noqa: E402


def bits(x):
    return struct.pack(">d", float(x)).hex()


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
    dev = torch.device("cuda", 0)
    loop = HotLoop(conf, dev, n_frames=64, H=512, W=512, curves=True)
    for it in range(iters):
        loss, rays = loop.step(it)
        torch.cuda.synchronize()
        info = loop.info
        terms = {k: v for k, v in info.items() if torch.is_tensor(v) and v.numel() == 1}
        terms.update({"fl_" + k: v for k, v in info.get("fl_loss", {}).items() if torch.is_tensor(v)})
        state = sum(float(p.detach().double().abs().sum()) for p in loop.shared_parameters())
        verts = sum(float(v.detach().double().abs().sum()) for v in loop.garment_vs)
        curves = sum(float(p.detach().double().abs().sum()) for p in loop.inter_free_curve.parameters())
        print("it %3d loss %s rays %d conv %s state %s verts %s curves %s" % (it, bits(loss), rays, info.get("rays_converged"),
                                                                        bits(state), bits(verts), bits(curves)))
        for k in sorted(terms):
            print("      %-28s %s" % (k, bits(terms[k])))
        grads = {"sdf0": loop.garment_nets[0].lin4.weight_v.grad, "sdf1": loop.garment_nets[1].lin4.weight_v.grad,
                 "def": loop.deformer.defs[0].lin2.weight.grad, "rend": loop.netRender.lin2.weight_v.grad,
                 "poses": loop.dataset.poses.grad, "dcond": loop.dataset.d_cond.grad, "focal": loop.dataset.focal.grad}
        for k, g in grads.items():
            if g is not None:
                print("      grad %-23s %s" % (k, bits(g.double().abs().sum())))
        for g_i, t in enumerate(loop.TmpPs):                # the surface points, their gradients, the jets the implicit differentiation reads
            if t is not None and t.grad is not None:
                pre = getattr(loop, "_prop_pre", {}).get(g_i)
                print("      TmpPs[%d] %-14s p %s grad %s jets %s %s" % (
                    g_i, tuple(t.shape), bits(t.detach().double().abs().sum()), bits(t.grad.double().abs().sum()),
                    bits(pre[1].double().abs().sum()) if pre is not None and pre[1] is not None else "-",
                    bits(pre[2].double().abs().sum()) if pre is not None and pre[2] is not None else "-"))
        if os.environ.get("RECMV_PROBE_ALL") == "1":      # every shared tensor's gradient, by position and shape
            named = {}
            for mod_name, mod in (("sdf0", loop.garment_nets[0]), ("sdf1", loop.garment_nets[1]), ("def", loop.deformer),
                                  ("rend", loop.netRender)):
                for n_, p_ in mod.named_parameters():
                    named[id(p_)] = mod_name + "." + n_
            for n_ in ("poses", "trans", "d_cond", "rendcond", "focal", "pp", "T"):
                t_ = getattr(loop.dataset, n_, None)
                if t_ is not None:
                    named[id(t_)] = "dataset." + n_
            for i, p_ in enumerate(loop.shared_parameters()):
                g_ = p_.grad
                print("      all %3d %-34s %-18s %s" % (i, named.get(id(p_), "?"), tuple(p_.shape),
                                                       bits(g_.double().abs().sum()) if g_ is not None else "-"))


if __name__ == "__main__":
    main()
