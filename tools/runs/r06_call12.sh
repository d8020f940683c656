#!/bin/bash
# round 6, GPU call 12: fused rotary backward / forward bit-equality after the contraction fix, ops + model suites, A/B timing of the decoder layer pieces
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c12; mkdir -p $O
python tools/_scratch_dbg_rope.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
timeout 1200 python -m pytest tests/test_ops_gpu.py -q 2>&1 | tail -6 | cut -c1-400
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_custom_ops_gpu.py tests/test_fullwidth_gpu.py -q 2>&1 | tail -6 | cut -c1-400
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
res = {}
for rnd in range(3):
    for fuse in (False, True):
        ops.GEMM_FUSE_ROPE = fuse
        for _ in range(3): ops.gemm_nt_rope(a, w, bias, cos, sin, S=S, rope_cols=rc, D=D)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.gemm_nt_rope(a, w, bias, cos, sin, S=S, rope_cols=rc, D=D)
        e1.record(); torch.cuda.synchronize()
        res.setdefault('fused' if fuse else 'gemm + rope', []).append(round(1000 * e0.elapsed_time(e1) / 20, 1))
print(json.dumps({'qkv projection + rotary embedding, us': res}))
PY
