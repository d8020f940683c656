"""single-shape GEMM driver for rocprofv3 PMC passes: python tools/one_gemm.py M N K variant reps"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_flamingo_amd import ops
M, N, K, v, reps = [int(x) for x in sys.argv[1:6]]
dev = torch.device("cuda")
a = (torch.rand((M, K), device=dev) * 2 - 1).to(torch.bfloat16)
b = (torch.rand((N, K), device=dev) * 2 - 1).to(torch.bfloat16)
c = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
ops.gemm_set_variant(v)
for _ in range(reps):
    ops.gemm_nt(a, b, out=c)
torch.cuda.synchronize()
