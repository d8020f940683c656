#!/bin/bash
# round 6, GPU call 14: lane-major rotary tables in the dQ epilogue - bit-equality tests, timing of the decoder backward fused / separate, per-kernel trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c14; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "fused_rope or schedules_bit_equal or rope" 2>&1 | tail -4 | cut -c1-400
python - <<'PY' 2>&1 | tail -3
import sys, json, torch
sys.path.insert(0, '.')
from audio_flamingo_amd import ops
dev = torch.device('cuda')
B, S, Hq, Hkv, D = 8, 1024, 28, 4, 128
qkv = (torch.randn((B * S, (Hq + 2 * Hkv) * D), device=dev) * 0.5).to(torch.bfloat16)
do = (torch.randn((B * S, Hq * D), device=dev) * 0.5).to(torch.bfloat16)
inv = 1.0 / (1e6 ** (torch.arange(0, D, 2, device=dev, dtype=torch.float32) / D))
fr = torch.arange(S, device=dev, dtype=torch.float32)[:, None] * inv[None]
emb = torch.cat([fr, fr], -1)
cos, sin = emb.cos().to(torch.bfloat16).contiguous(), emb.sin().to(torch.bfloat16).contiguous()
pos = torch.arange(S, device=dev, dtype=torch.int32).repeat(B).contiguous()
o, lse = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=True)
res = {}
for rnd in range(3):
    for name, fuse, pp in (("separate", False, None), ("fused, lane-major tables", True, None), ("fused, per-row table reads (pos given)", True, pos)):
        ops.ATTN_FUSE_ROPE_BWD = fuse
        for _ in range(3):
            ops.attn_bwd(qkv, o, do, lse, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=True, rope=(cos, sin, pp))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.attn_bwd(qkv, o, do, lse, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=True, rope=(cos, sin, pp))
        e1.record(); torch.cuda.synchronize()
        res.setdefault(name, []).append(round(1000 * e0.elapsed_time(e1) / 20, 1))
print(json.dumps({'decoder backward incl. rotary backward, us': res}))
PY
