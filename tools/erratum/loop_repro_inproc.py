"""Run-to-run reproducibility of ONE optimiser iteration, counted in one process: the loop is built once, warmed up (re-mesh +
one plain iteration), its whole state is snapshotted, and every repetition restores the snapshot and runs the same iteration
again — same inputs, same host random draws; whatever differs between repetitions comes from the schedule.  Per configuration
("cell") the digests of the results are counted; per result group too, so a cell that parts says WHERE.

    python tools/loop_repro_inproc.py [reps=50] [cells=all|name,name,...] [steps_per_rep=1]

Cells toggle, at run time: the matrix mode (recmv_set_gemm_mode), the second garment's render chain on a side stream
(RECMV_RENDER_STREAMS), how the jets zero their tangent rows (recmv_set_jet_fill: kernel / hipMemsetAsync), the waits on the
lazily built per-weight-version caches (RECMV_CACHE_EVENTS), joint / per-garment implicit differentiation (RECMV_PROP_JOINT),
the poison detector (RECMV_POISON=1: every workspace and output buffer pre-filled with NaN — a read of memory nobody wrote shows
as a non-finite count in EVERY repetition), and the one-stream order (RECMV_SERIAL=1).
"""
import copy
import hashlib
import os
import sys
import time
from collections import Counter
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import torch  # noqa: E402
from recmv import _lib as L  # noqa: E402
from recmv.hocon import ConfigFactory  # noqa: E402
from recmv.loop import HotLoop  # noqa: E402

ENV_KEYS = ("RECMV_RENDER_STREAMS", "RECMV_CACHE_EVENTS", "RECMV_PROP_JOINT", "RECMV_POISON", "RECMV_SERIAL", "RECMV_MERGE_JETS",
            "RECMV_B3_PRESPLIT")

# name -> (gemm mode, jet fill by kernel, env)
CELLS = {
    "f32 side-stream": (0, 1, {"RECMV_RENDER_STREAMS": "1"}),
    "f32 one-ray-stream": (0, 1, {"RECMV_RENDER_STREAMS": "0"}),
    "bf16x6 side-stream": (1, 1, {"RECMV_RENDER_STREAMS": "1"}),
    "bf16x6 one-ray-stream": (1, 1, {"RECMV_RENDER_STREAMS": "0"}),
    "bf16x6 side-stream memset-fill": (1, 0, {"RECMV_RENDER_STREAMS": "1"}),
    "bf16x6 side-stream memset-fill per-garment-prop": (1, 0, {"RECMV_RENDER_STREAMS": "1", "RECMV_PROP_JOINT": "0"}),
    "bf16x6 side-stream memset-fill no-cache-events": (1, 0, {"RECMV_RENDER_STREAMS": "1", "RECMV_CACHE_EVENTS": "0"}),
    "bf16x6 side-stream no-cache-events": (1, 1, {"RECMV_RENDER_STREAMS": "1", "RECMV_CACHE_EVENTS": "0"}),
    "f32 side-stream memset-fill no-cache-events": (0, 0, {"RECMV_RENDER_STREAMS": "1", "RECMV_CACHE_EVENTS": "0"}),
    "bf16x6 side-stream memset-fill POISON": (1, 0, {"RECMV_RENDER_STREAMS": "1", "RECMV_POISON": "1"}),
    "bf16x6 side-stream POISON": (1, 1, {"RECMV_RENDER_STREAMS": "1", "RECMV_POISON": "1"}),
    "f32 side-stream POISON": (0, 1, {"RECMV_RENDER_STREAMS": "1", "RECMV_POISON": "1"}),
    "bf16x6 side-stream no-presplit": (1, 1, {"RECMV_RENDER_STREAMS": "1", "RECMV_B3_PRESPLIT": "0"}),
    "bf16x6 side-stream two-jet-passes": (1, 1, {"RECMV_RENDER_STREAMS": "1", "RECMV_MERGE_JETS": "0"}),
    "bf16x6 side-stream no-curve-branch": (1, 1, {"RECMV_RENDER_STREAMS": "1", "REPRO_CURVES": "0"}),
    "bf16x6 one-ray-stream no-curve-branch": (1, 1, {"RECMV_RENDER_STREAMS": "0", "REPRO_CURVES": "0"}),
    "bf16x6 side-stream fam=big+mid (TN in f32)": (1, 1, {"RECMV_RENDER_STREAMS": "1", "REPRO_FAMILIES": "3"}),
    "bf16x6 side-stream fam=big+TN (mid NT in f32)": (1, 1, {"RECMV_RENDER_STREAMS": "1", "REPRO_FAMILIES": "5"}),
    "bf16x6 side-stream fam=mid+TN (big NT in f32)": (1, 1, {"RECMV_RENDER_STREAMS": "1", "REPRO_FAMILIES": "6"}),
    "bf16x6 side-stream fam=big": (1, 1, {"RECMV_RENDER_STREAMS": "1", "REPRO_FAMILIES": "1"}),
    "bf16x6 side-stream fam=mid": (1, 1, {"RECMV_RENDER_STREAMS": "1", "REPRO_FAMILIES": "2"}),
    "bf16x6 side-stream fam=TN": (1, 1, {"RECMV_RENDER_STREAMS": "1", "REPRO_FAMILIES": "4"}),
    "bf16x6 serial": (1, 1, {"RECMV_SERIAL": "1"}),
    "f32 serial": (0, 1, {"RECMV_SERIAL": "1"}),
}


def md5(*tensors):
    h = hashlib.md5()
    for t in tensors:
        if t is None:
            h.update(b"-")
        else:
            h.update(t.detach().contiguous().cpu().numpy().tobytes())
    return h.hexdigest()[:8]


class Snapshot:
    """Everything an iteration reads and changes: the shared tensors and their Adam state, the explicit vertices and their SGD
    state, the curves and their AdamW state, the counters, both random generators."""

    def __init__(self, loop):
        self.loop = loop
        self.tensors = [(p, p.detach().clone()) for p in self._all(loop)]
        self.opts = [(o, copy.deepcopy(o.state_dict())) for o in self._opts(loop)]
        self.counters = (loop.forward_time, loop.opt_times)
        self.cuda = torch.device(loop.device).type == "cuda"
        self.rng = (torch.get_rng_state(), torch.cuda.get_rng_state(loop.device) if self.cuda else None)

    @staticmethod
    def _all(loop):
        out = list(loop.shared_parameters()) + list(loop.garment_vs)
        if getattr(loop, "curves", False):
            out += list(loop.inter_free_curve.parameters())
        return out

    @staticmethod
    def _opts(loop):
        out = [loop.optimizer, loop.garment_optimizer]
        if getattr(loop, "curves", False):
            out.append(loop.fl_optimizer)
        return out

    def restore(self):
        loop = self.loop
        with torch.no_grad():
            for p, saved in self.tensors:
                p.copy_(saved)
                p.grad = None
        for o, sd in self.opts:
            o.load_state_dict(copy.deepcopy(sd))
        loop.forward_time, loop.opt_times = self.counters
        torch.set_rng_state(self.rng[0])
        if self.cuda:
            torch.cuda.set_rng_state(self.rng[1], loop.device)
            torch.cuda.synchronize()


def digests(loop, loss):
    named = {
        "loss": [loss],
        "sdf0.grad": [p.grad for p in loop.garment_nets[0].parameters()],
        "sdf1.grad": [p.grad for p in loop.garment_nets[1].parameters()] if loop.garment_size > 1 else [],
        "deformer.grad": [p.grad for p in loop.deformer.parameters()],
        "render.grad": [p.grad for p in loop.netRender.parameters()],
        "frames.grad": [p.grad for p in loop.dataset.learnable_weights()],
        "verts": list(loop.garment_vs),
        "curves": list(loop.inter_free_curve.parameters()) if getattr(loop, "curves", False) else [],
        "TmpPs": [t for t in loop.TmpPs if t is not None],
        "TmpPs.grad": [t.grad for t in loop.TmpPs if t is not None],
        "params_after": list(loop.shared_parameters()),
    }
    for n_, p_ in loop.deformer.named_parameters():          # which tensor of the offset MLP
        named["def:" + n_.replace("defs.0.", "")] = [p_.grad]
    for k_, v_ in loop.info.items():                         # every loss term the iteration reports
        if torch.is_tensor(v_) and v_.numel() == 1:
            named["info:" + k_] = [v_]
    for name_, y_, J_ in loop.__dict__.get("_jet_log", []):
        named[name_ + " y"] = [y_]
        named[name_ + " J"] = [J_]
    for stage, grads in getattr(loop, "_stage_grads", {}).items():      # ... and after which phase (clones taken on the main stream)
        for n_, g_ in grads.items():
            named["%s:%s" % (stage, n_)] = [g_]
    bad = 0
    for ts in named.values():
        for t in ts:
            if t is not None and t.is_floating_point():
                bad += int((~torch.isfinite(t)).sum())
    return {k: md5(*v) for k, v in named.items()}, bad


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    which = sys.argv[2] if len(sys.argv) > 2 else "all"
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    names = list(CELLS) if which == "all" else [n.strip() for n in which.split(",")]
    lib = L.lib()
    conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
    cpu = os.environ.get("RECMV_REPRO_CPU") == "1"          # logic smoke test of this tool on the host (tiny scene)
    dev = torch.device("cpu") if cpu else torch.device("cuda", 0)
    t0 = time.time()
    lib.recmv_set_gemm_mode(0)
    if cpu:
        from oracle import cpu_port                         # (tool smoke test only: the host stand-in of tests/)
        cpu_port.install()
        conf.put('train.sample_pix_num', 32)
        loop = HotLoop(conf, "cpu", n_frames=12, H=64, W=64, resolutions=[(9, 11, 7), (17, 21, 13)], skin_grid=(5, 9, 7), curves=True)
    else:
        loop = HotLoop(conf, dev, n_frames=64, H=512, W=512, curves=True)
    for it in range(2):                       # re-mesh + one plain iteration: every kernel loaded, every pool grown
        loop.step(it)
    if not cpu:
        torch.cuda.synchronize()
    snap = Snapshot(loop)
    if os.environ.get("RECMV_REPRO_STAGES") == "1":
        # the offset MLP's gradients as they stand after the mask loss's backward, after the final backward and after the implicit
        # differentiation: stream-ordered clones on the main stream, no host synchronisation added
        def grab(stage):
            loop.__dict__.setdefault("_stage_grads", {})[stage] = {
                n_.replace("defs.0.", ""): (p_.grad.clone() if p_.grad is not None else None)
                for n_, p_ in loop.deformer.named_parameters() if n_.endswith("lin0.weight") or n_.endswith("lin2.weight")
                or n_.endswith("lin4.bias")}
        orig_mask, orig_prop = loop.mask_loss, loop.propagateTmpPsGrad

        def mask_loss(*a, **k):
            out = orig_mask(*a, **k)
            grab("after_mask")
            return out

        def prop(*a, **k):
            grab("after_backward")
            return orig_prop(*a, **k)
        loop.mask_loss, loop.propagateTmpPsGrad = mask_loss, prop
    if os.environ.get("RECMV_REPRO_JETS") == "1":
        # every jet pass of the iteration (SDF nets and offset MLP, all streams): its outputs cloned on the stream that produced them
        from recmv import chains
        orig_jet = chains.mlp_jet
        jets = loop.__dict__.setdefault("_jet_log", [])

        def mlp_jet(x, cond, cond_index, Ws, bs, dims, *a, **k):
            y, J = orig_jet(x, cond, cond_index, Ws, bs, dims, *a, **k)
            jets.append(("jet%02d P=%d dims=%d..%d" % (len(jets), x.shape[0], dims[0], dims[-1]), y.detach().clone(), J.detach().clone()))
            return y, J
        chains.mlp_jet = mlp_jet
        from recmv import ops as _ops
        orig_regu = _ops.def_regu

        def def_regu(J, c):
            y = orig_regu(J, c)
            n_ = sum(1 for e in jets if e[0].startswith("regu"))
            jets.append(("regu%d P=%d" % (n_, J.shape[0]), y.detach().clone(), J.detach().clone()))
            return y
        _ops.def_regu = def_regu
        # ... and what the regulariser's kernel was handed (the contiguous copy of J) and what it left for the backward pass
        import ctypes as C_

        def regu_forward(ctx, J, c):
            Jc = J.detach().contiguous()
            P_ = Jc.shape[0]
            y = torch.empty(P_, dtype=torch.float32, device=J.device)
            gJ = torch.empty_like(Jc)
            L.check(L.lib().recmv_def_regu(L.ptr(Jc), P_, float(c), L.ptr(y), L.ptr(gJ), L.stream_ptr(J.device)), "def_regu")
            n_ = sum(1 for e in jets if e[0].startswith("kern"))
            jets.append(("kern%d regu input copy / dy/dJ" % n_, Jc.clone(), gJ.clone()))
            y2, gJ2 = torch.empty_like(y), torch.empty_like(gJ)           # the same launch once more, right behind the first
            L.check(L.lib().recmv_def_regu(L.ptr(Jc), P_, float(c), L.ptr(y2), L.ptr(gJ2), L.stream_ptr(J.device)), "def_regu")
            jets.append(("kern%d regu second launch y / dy/dJ" % n_, y2, gJ2))
            ctx.save_for_backward(gJ)
            return y
        _ops.DefRegu.forward = staticmethod(regu_forward)
    print("# built + warmed up in %.1f s; %d repetitions of %d iteration(s) per cell; vertices %s" % (
        time.time() - t0, reps, steps, [int(v.shape[0]) for v in loop.garment_vs]), flush=True)
    for name in names:
        mode, fill, env = CELLS[name]
        for k in ENV_KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        lib.recmv_set_gemm_mode(mode)
        lib.recmv_set_jet_fill(fill)
        lib.recmv_set_b3_families(int(os.environ.pop("REPRO_FAMILIES", "7")))
        loop.curves = os.environ.pop("REPRO_CURVES", "1") != "0"       # (the curve branch and its stream: a run-time switch of the loop)
        whole, groups, bad_total, rays = Counter(), {}, 0, None
        t1 = time.time()
        ref, reports, report = None, 0, []
        for _rep in range(reps):
            snap.restore()
            loop.__dict__.get("_jet_log", []).clear()
            for s in range(steps):
                loss, rays = loop.step(2 + s)
            if not cpu:
                torch.cuda.synchronize()
            d, bad = digests(loop, loss)
            bad_total += bad
            # where and by how much a repetition parts from the cell's first one (the offset MLP's weight gradients + the loss)
            cur = {"loss": loss.detach().double().view(1, 1).cpu()}
            for n_, p_ in loop.deformer.named_parameters():
                if n_.endswith("weight") and p_.grad is not None and "defs.0" in n_:
                    cur[n_.replace("defs.0.", "")] = p_.grad.detach().cpu()
            for k_, t_ in loop.__dict__.get("_stage_grads", {}).get("after_backward", {}).items():
                if t_ is not None and t_.dim() == 2:
                    cur["after_backward:" + k_] = t_.cpu()
            if ref is None:
                ref = cur
            elif reports < 6:
                lines = []
                for k_, t_ in cur.items():
                    r_ = ref[k_]
                    ne = (t_ != r_)
                    if bool(ne.any()):
                        rows, cols = ne.any(1).nonzero().view(-1), ne.any(0).nonzero().view(-1)
                        dd = (t_.double() - r_.double()).abs()
                        lines.append("      %-28s %d of %d elements differ, max |d| %.3e (max |ref| %.3e), rows %d..%d (%d distinct), cols %d..%d (%d distinct)" % (
                            k_, int(ne.sum()), ne.numel(), float(dd.max()), float(r_.double().abs().max()), int(rows.min()), int(rows.max()),
                            rows.numel(), int(cols.min()), int(cols.max()), cols.numel()))
                if lines:
                    reports += 1
                    report.append("    repetition %d parts from repetition 0:\n%s" % (_rep, "\n".join(lines)))
            whole[md5(*[torch.frombuffer(bytearray("".join(d.values()).encode()), dtype=torch.uint8)])] += 1
            for k, v in d.items():
                groups.setdefault(k, Counter())[v] += 1
        parted = {k: sorted(c.values(), reverse=True) for k, c in groups.items() if len(c) > 1}
        print("%-52s %s  non-finite %d  rays %s conv %s  %.1f s%s" % (
            name, " ".join("%s x%d" % kv for kv in whole.most_common()), bad_total, rays, loop.info.get("rays_converged"),
            time.time() - t1, ("   PARTED in: " + ", ".join("%s %s" % kv for kv in parted.items())) if parted else ""), flush=True)
        for r_ in report:
            print(r_, flush=True)
    for k in ENV_KEYS:
        os.environ.pop(k, None)
    lib.recmv_set_gemm_mode(0)
    lib.recmv_set_jet_fill(1)
    lib.recmv_set_b3_families(7)


if __name__ == "__main__":
    main()
