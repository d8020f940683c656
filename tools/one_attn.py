"""attention driver for rocprofv3 PMC passes: python tools/one_attn.py enc|dec|long5|long10 reps"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_flamingo_amd import ops
SHAPES = {"enc": (8, 1500, 20, 20, 64, False), "dec": (8, 1024, 28, 4, 128, True), "long5": (1, 7774, 28, 4, 128, True), "long10": (1, 15274, 28, 4, 128, True),
          "enc10": (20, 1500, 20, 20, 64, False)}
which, reps = sys.argv[1], int(sys.argv[2])
B, S, Hq, Hkv, D, causal = SHAPES[which]
dev = torch.device("cuda")
qkv = (torch.randn((B * S, (Hq + 2 * Hkv) * D), device=dev) * 0.5).to(torch.bfloat16)
do = (torch.randn((B * S, Hq * D), device=dev) * 0.5).to(torch.bfloat16)
for _ in range(reps):
    o, lse = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
    ops.attn_bwd(qkv, o, do, lse, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
torch.cuda.synchronize()
