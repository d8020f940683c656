"""timing-only diagnosis of the forward attention kernel (AFK_ATTN_DBG bits; results are WRONG by construction for dbg != 0)"""
import sys, os, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 1:
    for dbg in (0, 1, 2, 3, 4, 8, 6, 14, 15):
        env = dict(os.environ, AFK_ATTN_DBG=str(dbg))
        out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1]
        print(dbg, out, flush=True)
    sys.exit(0)
import torch
from audio_flamingo_amd import ops
dev = torch.device("cuda")
res = {}
for name, B, S, Hq, Hkv, D, causal in [("enc", 8, 1500, 20, 20, 64, False), ("dec", 8, 1024, 28, 4, 128, True), ("long", 1, 7774, 28, 4, 128, True)]:
    qkv = (torch.randn((B * S, (Hq + 2 * Hkv) * D), device=dev) * 0.5).to(torch.bfloat16)
    flops = 4.0 * B * Hq * S * S * D * (0.5 if causal else 1.0)
    for _ in range(2):
        ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    res[name] = (round(t * 1000), round(flops / t / 1e9))
print(json.dumps(res))
