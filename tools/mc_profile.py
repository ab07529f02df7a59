"""Run the 257^3 extraction (Seg3dLossless query + MC) a few times — wrap in rocprofv3 --kernel-trace --stats."""
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
for p in (REPO / "rec-mv_amd", REPO):
    sys.path.insert(0, str(p))
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
out = bench.mc_extract_timing(dev)
print(out)
