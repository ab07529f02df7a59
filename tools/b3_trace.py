"""Phase timeline of one workgroup of the staged bf16x6 GEMM (debug build path RECMV_B3_ABLATE=3)."""
import ctypes as C
import os
import sys
os.environ["RECMV_B3_ABLATE"] = "3"
sys.path.insert(0, 'rec-mv_amd')
import torch
from recmv import _lib as L, ops
dev = torch.device('cuda', 0)
M, N, K = 460800, 512, 512
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev) / K ** 0.5; bias = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev)
L.lib().recmv_set_gemm_mode(1)
buf = torch.zeros(4 * 64 * 8, dtype=torch.int64, device=dev)
fn = L.lib().recmv_debug_b3_trace
fn.restype = C.c_int
fn.argtypes = [C.c_void_p]
for _ in range(2):
    ops.gemm_nt(A, B, bias, 0, 100.0, 1.0, out=out)
fn(buf.data_ptr())
ops.gemm_nt(A, B, bias, 0, 100.0, 1.0, out=out)
torch.cuda.synchronize()
t = buf.cpu().view(4, 64, 8)
t0 = int(t[:, 0, 0].min())
names = ["top", "half", "mfma_end", "after_bar1", "after_lstore", "after_bar2"]
for w in range(4):
    print("wave", w)
    for kt in range(15):
        row = t[w, kt, :6].tolist()
        if row[0] == 0:
            continue
        rel = [r - t0 for r in row]
        d = [rel[i + 1] - rel[i] for i in range(5)]
        print(f"  kt {kt:2d} start {rel[0]:7d}  region1 {d[0]:5d}  region2 {d[1]:5d}  bar1 {d[2]:5d}  lstore {d[3]:5d}  bar2 {d[4]:5d}  total {rel[5]-rel[0]:6d}")
