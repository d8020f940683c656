"""left-padded causal attention at the decoder's shape: LDS-staged kernels with kv_lo vs the interval kernels (what such batches took before)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_flamingo_amd import ops
dev = torch.device("cuda")
B, S, Hq, Hkv, D = 8, 1024, 28, 4, 128
g = torch.Generator(device="cpu").manual_seed(0)
lo = torch.randint(0, 400, (B,), generator=g).to(torch.int32)
lo[0] = 0
qkv = (torch.randn((B * S, (Hq + 2 * Hkv) * D), device=dev) * 0.5).to(torch.bfloat16)
do = (torch.randn((B * S, Hq * D), device=dev) * 0.5).to(torch.bfloat16)
kv_lo = lo.to(dev)
i1 = torch.arange(1, S + 1, device=dev, dtype=torch.int32)[None]
krange = torch.stack([kv_lo[:, None].expand(B, S), torch.maximum(i1.expand(B, S), kv_lo[:, None])], -1).contiguous()
scale = D ** -0.5


def lds():
    o, lse = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=scale, causal=True, kv_lo=kv_lo)
    return o, ops.attn_bwd(qkv, o, do, lse, B, S, Hq, Hkv, D, scale=scale, causal=True, kv_lo=kv_lo)


def interval():
    o, lse = ops.attn_interval_fwd(qkv, krange, B, S, Hq, Hkv, D, scale=scale)
    return o, ops.attn_interval_bwd(qkv, o, do, lse, krange, B, S, Hq, Hkv, D, scale=scale)


res = {}
outs = {}
for name, fn in (("lds_kv_lo", lds), ("interval", interval)):
    for _ in range(2):
        outs[name] = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    res[name + "_fwd_bwd_ms"] = round(e0.elapsed_time(e1) / 5, 3)
valid = (torch.arange(S, device=dev)[None, :] >= kv_lo[:, None]).reshape(-1)
res["max_diff_o_valid_rows"] = float((outs["lds_kv_lo"][0][valid].float() - outs["interval"][0][valid].float()).abs().max())
res["max_diff_dqkv_valid_rows"] = float((outs["lds_kv_lo"][1][valid].float() - outs["interval"][1][valid].float()).abs().max())
res["kv_lo"] = lo.tolist()
print(json.dumps(res))
