import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_flamingo_amd import ops
dev = torch.device("cuda")
shapes = [(8192, 37888, 3584), (8192, 3584, 18944), (37888, 3584, 8192), (3584, 18944, 8192), (8192, 18944, 3584), (8192, 3584, 37888), (8192, 8192, 8192), (12000, 5120, 1280), (8192, 3584, 3584)]
cfgs = [("gm4", 2 + 256 * 4), ("gm2", 2 + 256 * 2), ("gm8", 2 + 256 * 8), ("gm16", 2 + 256 * 16), ("raw_gm4", 2 + 256 * (128 + 4))]
for M, N, K in shapes:
    a = (torch.rand((M, K), device=dev) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand((N, K), device=dev) * 2 - 1).to(torch.bfloat16)
    c = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    res = {}
    for rnd in range(2):   # interleaved rounds
        for name, code in cfgs:
            ops.gemm_set_variant(code)
            ops.gemm_nt(a, b, out=c)
            torch.cuda.synchronize()
            ops.prof_reset(); ops.prof_enable(True)
            for _ in range(4):
                ops.gemm_nt(a, b, out=c)
            ops.prof_enable(False)
            ms, fl, n = ops.prof_collect()
            res.setdefault(name, []).append(round(fl / ms / 1e9))
    ops.gemm_set_variant(0)
    print((M, N, K), res, flush=True)
