"""Does the packed-f32 fault (tools/packed_f32_bf16_mfma_repro.hip) also show beside torch's OWN matrix products — i.e. is it a
property of the platform rather than of this library's bf16x6 kernels?  The victim kernel (tools/packed_f32_victim_lib.hip: packed
f32 FMAs / adds, or the same arithmetic kept scalar) runs on the current stream while two side streams run torch.matmul in bf16 /
f16 / f32 (hipBLASLt / rocBLAS kernels); a launch is bad when its output differs from the first launch's.

    python tools/packed_f32_vs_torch_bf16.py [launches=200]
"""
import ctypes as C
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
lib = C.CDLL(str(REPO / "tools" / "bin" / "libpacked_victim.so"))
lib.launch_victim.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_void_p]
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
P = 30714
J = (torch.eye(3).view(1, 3, 3) + 0.06 * (torch.rand(P, 3, 3, generator=g) - 0.5)).to(dev).contiguous()
side = [torch.cuda.Stream(), torch.cuda.Stream()]
mats = {}
for name, dt in (("bf16", torch.bfloat16), ("f16", torch.float16), ("f32", torch.float32)):
    mats[name] = (torch.randn(12000, 4096, generator=g).to(dev).to(dt), torch.randn(4096, 512, generator=g).to(dev).to(dt))


def victim(out, paired):
    rc = lib.launch_victim(J.data_ptr(), P, out.data_ptr(), paired, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


for name in ("bf16", "f16", "f32"):
    a, b = mats[name]
    for paired in (1, 0):
        torch.cuda.synchronize()
        ref = torch.empty(P, 9, device=dev)
        victim(ref, paired)
        torch.cuda.synchronize()
        bad, lanes = 0, [0, 0, 0, 0]
        for _ in range(launches):
            for st in side:
                with torch.cuda.stream(st):
                    torch.matmul(a, b)
                    torch.matmul(a, b)
            out = torch.empty(P, 9, device=dev)
            victim(out, paired)
            ne = (out != ref).any(1)
            n = int(ne.sum())
            if n:
                bad += 1
                idx = ne.nonzero().view(-1) % 64
                for q in range(4):
                    lanes[q] += int(((idx // 16) == q).sum())
        print("torch.matmul %-4s 12000x4096x512 on two side streams, victim %-34s %3d of %d launches differ; wrong threads by quarter of the "
              "wave [0-15 | 16-31 | 32-47 | 48-63]: %s" % (name, "packed f32 (v_pk_fma/add/mul_f32)," if paired else "scalar FMAs / adds,", bad,
                                                            launches, lanes), flush=True)
