"""norm forward / backward microbenchmark at the training-step shapes (torch.cuda.Event timed, achieved HBM GB/s on algorithmic bytes:
forward 4 B/elem, backward 6 B/elem + 2 with the fused residual-gradient merge).  AFK_NORM_BWD=rows selects the row-per-wave backward,
AFK_NORM_BWD_R=4 the 4-row groups of the column-owned RMSNorm backward."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_flamingo_amd import ops

dev = torch.device("cuda")
BF = torch.bfloat16
out = {"AFK_NORM_BWD": os.environ.get("AFK_NORM_BWD", "cols"), "AFK_NORM_BWD_R": os.environ.get("AFK_NORM_BWD_R", "2")}
for kind, rows, D in (("rms", 8192, 3584), ("ln", 12000, 1280), ("rms", 7774, 3584), ("ln", 30000, 1280)):
    x = torch.randn((rows, D), device=dev).to(BF)
    dy = torch.randn((rows, D), device=dev).to(BF)
    skip = torch.randn((rows, D), device=dev).to(BF)
    w = torch.ones(D, device=dev, dtype=BF)
    b = torch.zeros(D, device=dev, dtype=BF)
    dw, db = torch.empty(D, device=dev, dtype=BF), torch.empty(D, device=dev, dtype=BF)
    if kind == "ln":
        y, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-5)
        fwd = lambda: ops.layernorm_fwd(x, w, b, 1e-5)
        bwd = lambda: ops.layernorm_bwd(x, w, dy, mean, rstd, dw, db, dx_add=skip)
    else:
        y, rstd = ops.rmsnorm_fwd(x, w, 1e-6)
        fwd = lambda: ops.rmsnorm_fwd(x, w, 1e-6)
        bwd = lambda: ops.rmsnorm_bwd(x, w, dy, rstd, dw, dx_add=skip)
    res = {}
    for name, fn, bpe in (("fwd", fwd, 4), ("bwd+skip", bwd, 8)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        res[name] = {"us": round(us, 1), "TB/s": round(rows * D * bpe / us / 1e6, 2)}
    out[f"{kind} {rows}x{D}"] = res
print(json.dumps(out))
