"""A few launches of the row-tile MLP passes at one size, for counter / trace runs:  python tools/mlp_rows_one.py P row_tiles [reps]"""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO), str(REPO / "tests" / "golden")]
import torch  # noqa: E402
import common_setup as cs  # noqa: E402

P, rt = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
import recmv.chains as chains  # noqa: E402
from recmv import _lib as L  # noqa: E402
from recmv.model import getTmpSdf  # noqa: E402
chains.MLP_ROWS_MIN, chains.MLP_ROWS_MAX = 1, 1 << 20
L.lib().recmv_set_mlp_rows_tile(rt)
sdf = cs.build_sdf(getTmpSdf).to("cuda:0")
x = (torch.rand(P, 3, device="cuda:0") - 0.5) * 1.4
ch = sdf.chain(sdf._pe_weights({"sdfRatio": 0.8}), need_t=True)
for _ in range(reps):
    ch.forward(x, n_out=1, keep=True, slot="b")
    ch.vjp_input(x, None, slot="b")
torch.cuda.synchronize()
