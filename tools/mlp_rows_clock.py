"""Where a row-tile forward pass spends its time: build with  RECMV_HIPCC_EXTRA=-DRECMV_ROWS_TIMING python rec-mv_amd/build.py --force,
then  python tools/mlp_rows_clock.py P row_tiles .  Thread 0 of workgroup 0 stamps the 100 MHz clock at: start, input tile
built, then per hidden layer: products done (wave 0), epilogue done (wave 0), workgroup barrier passed, activations kept."""
import ctypes as C
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO), str(REPO / "tests" / "golden")]
import torch  # noqa: E402
import common_setup as cs  # noqa: E402

P, rt = int(sys.argv[1]), int(sys.argv[2])
import recmv.chains as chains  # noqa: E402
from recmv import _lib as L  # noqa: E402
from recmv.model import getTmpSdf  # noqa: E402
chains.MLP_ROWS_MIN, chains.MLP_ROWS_MAX = 1, 1 << 20
lib = L.lib()
lib.recmv_set_mlp_rows_tile(rt)
sdf = cs.build_sdf(getTmpSdf).to("cuda:0")
x = (torch.rand(P, 3, device="cuda:0") - 0.5) * 1.4
ch = sdf.chain(sdf._pe_weights({"sdfRatio": 0.8}), need_t=True)
buf = (C.c_longlong * 512)()
fn = lib.recmv_debug_rows_clock
fn.argtypes = [C.c_void_p, C.c_int, C.c_int]
for rep in range(3):
    ch.forward(x, n_out=1, keep=True, slot="b")
    n = fn(buf, 512, 1)
    t = [buf[i] for i in range(n)]
    d = [(b - a) * 10 for a, b in zip(t, t[1:])]         # ns
    print("rep", rep, "stamps", n, "total %.1f us" % ((t[-1] - t[0]) / 100.0))
    print("  input tile: %d ns" % d[0])
    k = 1
    lay = 0
    while k + 2 < len(d):
        print("  layer %d: products %6d ns   epilogue %5d ns   barrier %5d ns   keep %5d ns" % (lay, d[k], d[k + 1], d[k + 2], d[k + 3] if k + 3 < len(d) else -1))
        k += 4
        lay += 1
