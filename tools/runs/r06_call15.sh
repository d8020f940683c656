#!/bin/bash
# round 6, GPU call 15: lane-major tables in the qkv GEMM's RoPE epilogue - tests + timing; ops / model suites
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "rope" 2>&1 | tail -4 | cut -c1-400
python - <<'PY' 2>&1 | tail -3
import sys, json, torch
sys.path.insert(0, '.')
from audio_flamingo_amd import ops
dev = torch.device('cuda')
M, S, Hq, Hkv, D, K = 8192, 1024, 28, 4, 128, 3584
N, rc = (Hq + 2 * Hkv) * D, (Hq + Hkv) * D
a = torch.randn((M, K), device=dev).to(torch.bfloat16); w = (torch.randn((N, K), device=dev) * K ** -0.5).to(torch.bfloat16); bias = torch.randn(N, device=dev).to(torch.bfloat16)
inv = 1.0 / (1e6 ** (torch.arange(0, D, 2, device=dev, dtype=torch.float32) / D))
fr = torch.arange(S, device=dev, dtype=torch.float32)[:, None] * inv[None]
emb = torch.cat([fr, fr], -1); cos, sin = emb.cos().to(torch.bfloat16).contiguous(), emb.sin().to(torch.bfloat16).contiguous()
pos = torch.arange(S, device=dev, dtype=torch.int32).repeat(M // S).contiguous()
res = {}
for rnd in range(3):
    for name, fuse, pp in (("gemm + rope", False, None), ("fused, lane-major tables", True, None), ("fused, per-row table reads (pos given)", True, pos), ("plain gemm (no rope)", None, None)):
        def run():
            if fuse is None: return ops.gemm_nt(a, w, bias=bias)
            ops.GEMM_FUSE_ROPE = fuse
            return ops.gemm_nt_rope(a, w, bias, cos, sin, S=S, rope_cols=rc, D=D, pos=pp)
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        res.setdefault(name, []).append(round(1000 * e0.elapsed_time(e1) / 20, 1))
print(json.dumps({'qkv projection + rotary embedding, us': res}))
PY
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_custom_ops_gpu.py -q 2>&1 | tail -4 | cut -c1-300
