import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_flamingo_amd import ops
dev = torch.device("cuda")
for name, B, S, Hq, Hkv, D, causal in [("enc", 8, 1500, 20, 20, 64, False), ("long", 1, 7774, 28, 4, 128, True)]:
    qkv = (torch.randn((B * S, (Hq + 2 * Hkv) * D), device=dev) * 0.5).to(torch.bfloat16)
    for _ in range(3):
        o, lse = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
    torch.cuda.synchronize()
    c = lse[0, 0, :S].cpu()
    print(os.environ.get("AFK_ATTN_DBG"), name, "cycles/tile-iteration (clock64 = 100 MHz ticks?) first q-block", float(c[0]), "mid", float(c[S // 2]), "last", float(c[S - 1]), flush=True)
