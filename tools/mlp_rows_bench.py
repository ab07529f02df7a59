"""Row-tile fused MLP passes (csrc/mlp_rows.hip) against the per-layer launch chains (csrc/mlp_chain.hip) at the ray path's row counts:
SDF value + input gradient, offset MLP + its VJP — each pass sequence captured in a hipGraph (no Python between launches), 20 per
graph, replayed 5 times between two HIP events.   python tools/mlp_rows_bench.py > gpurun_out/r04_mlp_rows_bench.txt"""
import os
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO), str(REPO / "tests" / "golden")]
import torch  # noqa: E402

import bench  # noqa: E402
import common_setup as cs  # noqa: E402

DEV = "cuda:0"
RATIO = {"sdfRatio": 0.8, "deformerRatio": 0.7, "renderRatio": 1.0}


def main():
    from recmv.model import MLPTranslator, getTmpSdf
    sdf = cs.build_sdf(getTmpSdf).to(DEV)
    tr = cs.build_translator(MLPTranslator).to(DEV)
    conds = torch.randn(3, 128, device=DEV) * 0.1
    flop_sdf = 2 * sum(a * b for a, b in zip([39, 512, 512, 512, 512, 512, 512, 512], [512, 512, 512, 473, 512, 512, 512, 512])) + 2 * 512
    flop_tr = 2 * (167 * 512 + 3 * 512 * 512 + 512 * 3)
    print("# us per pass pair (forward keep + vjp_input); TFLOP/s counts 2x the forward FLOP for the pair")
    for P in (1024, 1500, 2048, 2560, 3072, 4096, 5000, 6144, 8192):
        x = (torch.rand(P, 3, device=DEV) - 0.5) * 1.4
        frame = torch.randint(0, 3, (P,), device=DEV)
        g3 = torch.randn(P, 3, device=DEV)
        row = ["P=%5d" % P]
        for rows in (1, 2, 0):
            os.environ["RECMV_MLP_ROWS"] = str(min(rows, 1))
            if rows:
                from recmv import _lib as L
                L.lib().recmv_set_mlp_rows_tile(rows)
            os.environ["RECMV_MLP_ROWS_MAX"] = "100000"
            import recmv.chains as chains
            chains.MLP_ROWS_MAX, chains.MLP_ROWS_MIN = 100000, 1
            ch = sdf.chain(sdf._pe_weights(RATIO), need_t=True)
            ct = tr.prepare_explicit(conds, ratio=RATIO)

            def sdf_pair():
                ch.forward(x, n_out=1, keep=True, slot="b")
                ch.vjp_input(x, None, slot="b")

            def tr_pair():
                ct.forward(x, cond=conds, cond_index=frame, n_out=3, keep=True, slot="b")
                ct.vjp_input(x, g3, slot="b")

            for name, fn, fl in (("sdf", sdf_pair, flop_sdf), ("offset", tr_pair, flop_tr)):
                sec, how = bench._graph_time(fn)
                row.append("%s[%s] %7.1f us %5.1f TF/s" % (name, ("rows%d" % (16 * rows)) if rows else "chain", sec * 1e6, 2 * fl * P / sec / 1e12))
        print("   ".join(row), flush=True)


if __name__ == "__main__":
    main()
