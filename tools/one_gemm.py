"""single-shape GEMM driver for rocprofv3 PMC passes: python tools/one_gemm.py M N K variant reps [form NT|NN|TN]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_flamingo_amd import ops
M, N, K, v, reps = [int(x) for x in sys.argv[1:6]]
form = sys.argv[6] if len(sys.argv) > 6 else "NT"
dev = torch.device("cuda")
r = lambda *s: (torch.rand(s, device=dev) * 2 - 1).to(torch.bfloat16)
c = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
ops.gemm_set_variant(v)
if form == "NT":
    a, b, kw = r(M, K), r(N, K), {}
elif form == "NN":
    a, b, kw = r(M, K), r(K, N), dict(trans_b=True)
else:
    a, b, kw = r(K, M), r(K, N), dict(trans_a=True, trans_b=True)
for _ in range(reps):
    ops.gemm(a, b, out=c, **kw)
torch.cuda.synchronize()
