"""GELU backward of the encoder's fc1 ([12000, 5120] at B = 8): elementwise kernel + column-sum pass against the column-owned fused form (AFK_GELU_CS_PARTS = row parts)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_flamingo_amd import ops
dev = torch.device("cuda")
rows, C = int(os.environ.get("ROWS", "12000")), int(os.environ.get("COLS", "5120"))
g = torch.Generator(device=dev).manual_seed(0)
NSET = 6
dy = [torch.randn(rows, C, device=dev, generator=g).bfloat16() for _ in range(NSET)]
pre = [torch.randn(rows, C, device=dev, generator=g).bfloat16() for _ in range(NSET)]
out = torch.empty(C, device=dev, dtype=torch.bfloat16)
def t(fn, iters=60):
    for i in range(NSET): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fn(i % NSET)
    e1.record(); torch.cuda.synchronize()
    return round(1e3 * e0.elapsed_time(e1) / iters, 1)
res = {"rows": rows, "cols": C, "parts": os.environ.get("AFK_GELU_CS_PARTS", "256")}
res["gelu_bwd_us"] = t(lambda i: ops.gelu_bwd(dy[i], pre[i]))
dx = ops.gelu_bwd(dy[0], pre[0])
res["colsum_us"] = t(lambda i: ops.colsum(dx, out))
res["gelu_bwd_colsum_us"] = t(lambda i: ops.gelu_bwd(dy[i], pre[i], colsum_out=out))
print(json.dumps(res))
