"""attention microbenchmark on the two AF3 shapes; both implementations ("lds" vs "direct"), torch.cuda.Event timed"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_flamingo_amd import ops
dev = torch.device("cuda")
for name, B, S, Hq, Hkv, D, causal in [("encoder", 8, 1500, 20, 20, 64, False), ("decoder", 8, 1024, 28, 4, 128, True)]:
    qkv = (torch.randn((B * S, (Hq + 2 * Hkv) * D), device=dev) * 0.5).to(torch.bfloat16)
    do = (torch.randn((B * S, Hq * D), device=dev) * 0.5).to(torch.bfloat16)
    flops_fwd = 4.0 * B * Hq * S * S * D * (0.5 if causal else 1.0)
    row = {"shape": name}
    outs = {}
    for impl in ("direct", "lds"):
        ops.ATTN_IMPL = impl
        o, lse = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
        dq = ops.attn_bwd(qkv, o, do, lse, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
        outs[impl] = (o, dq)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        for _ in range(5):
            o, lse = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
        e[1].record()
        for _ in range(5):
            ops.attn_bwd(qkv, o, do, lse, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
        e[2].record()
        torch.cuda.synchronize()
        tf, tb = e[0].elapsed_time(e[1]) / 5, e[1].elapsed_time(e[2]) / 5
        row[impl] = {"fwd_ms": round(tf, 3), "bwd_ms": round(tb, 3), "fwd_tflops": round(flops_fwd / tf / 1e9, 1), "bwd_tflops_alg2.5x": round(2.5 * flops_fwd / tb / 1e9, 1)}
    row["max_diff_o"] = float((outs["lds"][0].float() - outs["direct"][0].float()).abs().max())
    row["max_diff_dqkv"] = float((outs["lds"][1].float() - outs["direct"][1].float()).abs().max())
    print(json.dumps(row), flush=True)
