"""Which lines of the host code issue the most torch launches in one iteration of the hot loop ON THE DEVICE (the HIP entry points go
through ctypes and are not ATen calls: what is counted here is the torch glue between them).  Forward: every ATen call that launches
something is attributed to the innermost Python frame inside rec-mv_amd/recmv.  Backward: every autograd node created by a torch call
gets a pre-hook carrying the source line of that call, so the ATen calls the engine makes while it runs that node are attributed to
the line that built it.      python tools/op_census.py [--top 50] [--iters 3]"""
import argparse
import collections
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO), str(REPO / "tests" / "golden")]
import torch  # noqa: E402
from torch.overrides import TorchFunctionMode  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

PKG = str(REPO / "rec-mv_amd" / "recmv")
SKIP = ("aten.view", "aten._unsafe_view", "aten.t.", "aten.transpose", "aten.permute", "aten.expand", "aten.unsqueeze", "aten.squeeze",
        "aten.select", "aten.slice", "aten.alias", "aten.detach", "aten.as_strided", "aten.reshape", "aten.unbind", "aten.split",
        "aten.empty", "aten.sym_", "aten.is_", "aten._local_scalar_dense", "aten.lift_fresh", "aten.diagonal", "aten.narrow",
        "prim.", "aten.stride", "aten.size", "aten.numel", "aten.result_type", "aten.chunk", "aten.unfold", "aten._reshape_alias",
        "aten.set_", "aten.resize_", "aten.new_empty", "aten.empty_like", "aten.record_stream", "aten.new_empty_strided",
        "aten.empty_strided")      # views / metadata: no launch

STATE = {"line": None, "in_backward": False, "phase": "?"}
FWD = collections.Counter()
BWD = collections.Counter()
PHASE = collections.Counter()


def _here():
    f = sys._getframe(2)
    while f is not None:
        fn = f.f_code.co_filename
        if fn.startswith(PKG):
            return "%s:%d %s" % (fn[len(PKG) + 1:], f.f_lineno, f.f_code.co_name)
        f = f.f_back
    return None


class Hooker(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        if not STATE["in_backward"]:
            where = _here()
            if where is not None:
                outs = out if isinstance(out, (tuple, list)) else (out,)
                for o in outs:
                    if isinstance(o, torch.Tensor) and o.grad_fn is not None and not getattr(o.grad_fn, "_census", False):
                        try:
                            o.grad_fn.register_prehook(lambda g, w=where: STATE.__setitem__("line", w))
                        except Exception:
                            pass
        return out


class Counter(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            if STATE["in_backward"]:
                f, own = sys._getframe(0), None
                while f is not None and f is not STATE.get("engine_frame"):      # a custom Function's backward: its own frame
                    fn = f.f_code.co_filename
                    if fn.startswith(PKG):
                        own = "%s:%d %s" % (fn[len(PKG) + 1:], f.f_lineno, f.f_code.co_name)
                        break
                    f = f.f_back
                BWD[own or STATE["line"] or "(node built outside recmv/)"] += 1
                PHASE[(STATE["phase"], "backward")] += 1
            else:
                where = _here()
                FWD[where or "(outside recmv/)"] += 1
                PHASE[(STATE["phase"], "forward")] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=50)
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    import bench
    from recmv.hocon import ConfigFactory
    from recmv.loop import HotLoop
    dev = torch.device("cuda", 0)
    conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
    loop = HotLoop(conf, dev, stage="coarse", curves=True, **bench.HOTLOOP_KW)
    it0 = bench.load_scene(loop, bench.SCENE_FILE)       # the bench's frozen scene, right after its re-mesh
    for i in range(it0, it0 + 3):
        loop.step(i)
    torch.cuda.synchronize()
    import torch.autograd.graph as G
    real_engine = G._engine_run_backward

    def engine(*x, **k):
        prev = STATE["in_backward"]
        STATE["in_backward"], STATE["line"] = True, None
        STATE["engine_frame"] = sys._getframe()
        try:
            return real_engine(*x, **k)
        finally:
            STATE["in_backward"] = prev
    G._engine_run_backward = engine
    torch.autograd._engine_run_backward = engine
    import torch.autograd as TA
    TA._engine_run_backward = engine
    with Hooker(), Counter():
        for i in range(it0 + 3, it0 + 3 + a.iters):
            loop.step(i)
    torch.cuda.synchronize()
    n = float(a.iters)
    tf, tb = sum(FWD.values()) / n, sum(BWD.values()) / n
    print("# torch launches per iteration (ATen calls that launch; HIP entry points not included): forward / host code %.0f, backward engine %.0f" % (tf, tb))
    both = collections.Counter()
    for k, v in FWD.items():
        both[k] += v
    for k, v in BWD.items():
        both[k] += v
    print("# --- by source line, forward + the backward of the nodes it built (per iteration: total = forward + backward)")
    for k, v in both.most_common(a.top):
        print("%7.1f = %6.1f + %6.1f   %s" % (v / n, FWD.get(k, 0) / n, BWD.get(k, 0) / n, k))
    byfn = collections.Counter()
    for k, v in both.items():
        parts = k.split(" ")
        byfn[(parts[0].split(":")[0] + " " + parts[-1]) if ":" in parts[0] else k] += v
    print("# --- by function")
    for k, v in byfn.most_common(40):
        print("%7.1f   %s" % (v / n, k))


if __name__ == "__main__":
    main()
