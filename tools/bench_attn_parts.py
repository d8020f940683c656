"""GQA dK/dV sweep: number of partials per kv head (afk_attn_set_dkdv_parts; round 6).  P = group (7 for AF3's 28:4) is the one-block-per-query-head form of
rounds 2-5; smaller P = fewer, longer blocks and fewer partials through HBM.  HIP-event timing of the whole backward (dQ + dK/dV + reduce), alternating rounds,
and the deviation of dQKV from the P = group result."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_flamingo_amd import ops, _lib
dev = torch.device("cuda")
for name, B, S, Hq, Hkv, D in [("decoder S=1024", 8, 1024, 28, 4, 128), ("decoder S=2048", 4, 2048, 28, 4, 128), ("5-min decoder S=7774", 1, 7774, 28, 4, 128)]:
    qkv = (torch.randn((B * S, (Hq + 2 * Hkv) * D), device=dev) * 0.5).to(torch.bfloat16)
    do = (torch.randn((B * S, Hq * D), device=dev) * 0.5).to(torch.bfloat16)
    o, lse = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=True)
    group = Hq // Hkv
    parts = [group, 4, 3, 2, 1]
    ref = None
    row = {"shape": name, "us": {}, "max_abs_diff_vs_P7": {}}
    n = 20 if S < 4000 else 6
    for rnd in range(3):
        for P in parts:
            _lib.call("afk_attn_set_dkdv_parts", P)
            for _ in range(2):
                d = ops.attn_bwd(qkv, o, do, lse, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                d = ops.attn_bwd(qkv, o, do, lse, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=True)
            e1.record()
            torch.cuda.synchronize()
            row["us"].setdefault(f"P{P}", []).append(round(1000 * e0.elapsed_time(e1) / n, 1))
            if rnd == 0:
                if ref is None:
                    ref = d.clone()
                row["max_abs_diff_vs_P7"][f"P{P}"] = float((d.float() - ref.float()).abs().max())
    print(json.dumps(row), flush=True)
_lib.call("afk_attn_set_dkdv_parts", 0)
