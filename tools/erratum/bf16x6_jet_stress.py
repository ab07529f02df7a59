"""Is the offset MLP's jet pass (forward + backward, csrc/mlp_jet.hip) bit-reproducible when two of them run beside each other on two
streams — the situation of the two garments' render-loss terms — in the f32 and in the bf16x6 matrix mode?  The same pass is repeated on
fixed inputs on stream A while stream B runs its own; every repetition is compared bit for bit with the first.
    python tools/bf16x6_jet_stress.py [reps]"""
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO), str(REPO / "tests" / "golden")]
import torch  # noqa: E402
import common_setup as cs  # noqa: E402
from recmv import _lib as L  # noqa: E402
from recmv.model import MLPTranslator, getTmpSdf  # noqa: E402

dev = "cuda:0"
RATIO = {"sdfRatio": 1.0, "deformerRatio": 0.5, "renderRatio": 1.0}


def make_case(seed, P_pts, P_rays):
    g = torch.Generator().manual_seed(seed)
    pts = ((torch.rand(3, P_pts // 3, 3, generator=g) - 0.5) * 1.4).to(dev).requires_grad_(True)
    ps = ((torch.rand(P_rays, 3, generator=g) - 0.5) * 1.4).to(dev).requires_grad_(True)
    cond = (torch.randn(3, 128, generator=g) * 0.1).to(dev).requires_grad_(True)
    frame = torch.randint(0, 3, (P_rays,), generator=g).to(dev)
    w1 = torch.randn(3, P_pts // 3, 3, generator=g).to(dev)
    w2 = torch.randn(P_pts // 3 * 3, 3, 3, generator=g).to(dev)
    return pts, ps, cond, frame, w1, w2


def run_translator(tr, case):
    from recmv import utils
    pts, ps, cond, frame, w1, w2 = case
    for t in (pts, ps, cond):
        t.grad = None
    tr.zero_grad(set_to_none=True)
    out = tr.jet_two_blocks(pts, cond, ps, frame, RATIO["deformerRatio"], "x")
    J = utils.compute_Jacobian(pts, out, True, True)
    out2 = tr(ps, cond, frame, ratio=RATIO, offset_type="x", jet=True)
    J2 = utils.compute_Jacobian(ps, out2, True, True)
    loss = (out * w1).sum() + (J.reshape(-1, 3, 3) * w2).sum() + out2.sum() + (J2 * J2).sum()
    loss.backward()
    res = [out.detach(), J.detach(), out2.detach(), J2.detach(), pts.grad, ps.grad, cond.grad] + [p.grad for p in tr.parameters()]
    return [r.clone() for r in res]


def run_sdf(net, case):
    pts, ps, cond, frame, w1, w2 = case
    x = ps
    x.grad = None
    net.zero_grad(set_to_none=True)
    pred = net(x, RATIO, jet=True)
    gr = net.gradient(x, pred)
    loss = (pred[:, :1] ** 2).sum() + ((gr.norm(dim=1) - 1) ** 2).sum() + pred[:, 1:].sum() * 1e-3
    loss.backward()
    res = [pred.detach(), gr.detach(), x.grad] + [p.grad for p in net.parameters() if p.grad is not None]
    return [r.clone() for r in res]


def stress(label, fn_a, fn_b, reps):
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        ref = fn_a()
    torch.cuda.synchronize()
    bad, which = 0, {}
    for r in range(reps):
        with torch.cuda.stream(sb):
            fn_b()
            fn_b()
        with torch.cuda.stream(sa):
            res = fn_a()
        torch.cuda.synchronize()
        diff = [i for i, (a, b) in enumerate(zip(ref, res)) if not torch.equal(a, b)]
        if diff:
            bad += 1
            for i in diff:
                which[i] = which.get(i, 0) + 1
    print("%-58s %d of %d repetitions differ from the first %s" % (label, bad, reps, ("(result index: times) %s" % which) if bad else ""),
          flush=True)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    for mode in (0, 1):
        L.lib().recmv_set_gemm_mode(mode)
        tag = "f32   " if mode == 0 else "bf16x6"
        tr_a, tr_b = cs.build_translator(MLPTranslator).to(dev), cs.build_translator(MLPTranslator).to(dev)
        ca, cb = make_case(1, 6144, 3157), make_case(2, 6144, 3111)
        stress(tag + " translator jets, another translator pass beside it", lambda: run_translator(tr_a, ca), lambda: run_translator(tr_b, cb), reps)
        stress(tag + " translator jets, the SAME module beside it (other rows)", lambda: run_translator(tr_a, ca), lambda: run_translator(tr_a, cb), reps)
        sdf_a, sdf_b = cs.build_sdf(getTmpSdf).to(dev), cs.build_sdf(getTmpSdf).to(dev)
        stress(tag + " SDF jets, another net beside it", lambda: run_sdf(sdf_a, make_case(3, 6144, 9000)), lambda: run_sdf(sdf_b, make_case(4, 6144, 9000)), max(reps // 3, 20))
    L.lib().recmv_set_gemm_mode(0)


if __name__ == "__main__":
    main()
