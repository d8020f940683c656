"""per-kernel time of the five launches of a single-sequence decode layer (csrc/decode_chain.hip + afk_attn_decode_fused) at the AF3-7B widths, weights
rotated over NSET distinct sets so that nothing is served from the caches.   python tools/bench_decode_chain.py [keys]   (AFK_CHAIN_S / AFK_ATTN_DECODE_SYNC
are read by the library once per process)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_flamingo_amd import _lib, ops

dev = torch.device("cuda")
BF = torch.bfloat16
H, Hq, Hkv, D, I = 3584, 28, 4, 128, 18944
nq, nk = Hq * D, Hkv * D
keys = int(sys.argv[1]) if len(sys.argv) > 1 else 800
NSET, ITERS = 6, int(os.environ.get("ITERS", "60"))
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=0.02: (torch.randn(*s, device=dev, generator=g) * sc).to(BF)
W = [dict(qkv=rnd(nq + 2 * nk, H), o=rnd(H, nq), gu=rnd(2 * I, H), d=rnd(H, I)) for _ in range(NSET)]
x, x2, act, q, o = rnd(1, H, sc=1.0), torch.empty(1, H, device=dev, dtype=BF), torch.empty(1, I, device=dev, dtype=BF), torch.empty(1, nq, device=dev, dtype=BF), rnd(1, nq, sc=1.0)
nw, bias = rnd(H, sc=1.0), rnd(nq + 2 * nk, sc=0.1)
Smax = max(1024, keys + 64)
spad = ops.pad64(Smax)
Kc = [rnd(1, Smax, nk, sc=1.0) for _ in range(NSET)]
Vt = [rnd(1, Hkv, D, spad, sc=1.0) for _ in range(NSET)]
cos, sin = rnd(keys + 64, D, sc=1.0), rnd(keys + 64, D, sc=1.0)
pos = torch.tensor([keys], device=dev, dtype=torch.int32)
start = torch.tensor([keys], device=dev, dtype=torch.int32)
kr = torch.tensor([[0, keys + 1]], device=dev, dtype=torch.int32)
ns = int(os.environ.get("NS", "8"))
aws = torch.zeros(_lib.load().afk_attn_decode_workspace_floats(1, Hq, D, ns), device=dev, dtype=torch.float32)
st = ops._stream()
actr = rnd(1, I, sc=1.0)

def k_qkv(i): _lib.call("afk_decode_chain_qkv", x.data_ptr(), nw.data_ptr(), 1e-6, W[i]["qkv"].data_ptr(), H, H, bias.data_ptr(), cos.data_ptr(), sin.data_ptr(), pos.data_ptr(), q.data_ptr(), Kc[i].data_ptr(), Vt[i].data_ptr(), spad, start.data_ptr(), Hq, Hkv, D, st)
def k_attn(i): _lib.call("afk_attn_decode_fused", o.data_ptr(), nq, D, Kc[i].data_ptr(), Smax * nk, nk, D, Vt[i].data_ptr(), Hkv * D * spad, spad, q.data_ptr(), nq, D, kr.data_ptr(), 1, Hq, Hkv, D, float(D ** -0.5), ns, aws.data_ptr(), st)
def k_attn2(i): _lib.call("afk_attn_decode", o.data_ptr(), nq, D, Kc[i].data_ptr(), Smax * nk, nk, D, Vt[i].data_ptr(), Hkv * D * spad, spad, q.data_ptr(), nq, D, kr.data_ptr(), 1, Hq, Hkv, D, float(D ** -0.5), ns, aws.data_ptr(), st)
def k_o(i): _lib.call("afk_decode_chain_linear_residual", o.data_ptr(), W[i]["o"].data_ptr(), nq, H, nq, x.data_ptr(), x2.data_ptr(), st)
def k_gu(i): _lib.call("afk_decode_chain_gate_up", x.data_ptr(), nw.data_ptr(), 1e-6, W[i]["gu"].data_ptr(), H, I, H, act.data_ptr(), st)
Vv = 152064
Wh = [rnd(Vv, H) for _ in range(2)]
logits = torch.empty(1, Vv, device=dev, dtype=torch.float32)
def k_head(i): _lib.call("afk_decode_chain_lm_head", x.data_ptr(), nw.data_ptr(), 1e-6, Wh[i % 2].data_ptr(), H, Vv, H, logits.data_ptr(), None, None, st)
def k_d(i): _lib.call("afk_decode_chain_linear_residual", actr.data_ptr(), W[i]["d"].data_ptr(), I, H, I, x.data_ptr(), x2.data_ptr(), st)

bytes_ = dict(qkv=2.0 * (nq + 2 * nk) * H, attn=2.0 * 2 * (keys + 1) * nk, attn_two_launches=2.0 * 2 * (keys + 1) * nk, o_proj=2.0 * H * nq, gate_up=2.0 * 2 * I * H, down=2.0 * H * I, lm_head=2.0 * Vv * H)
res = {"keys": keys, "nsplit": ns, "AFK_CHAIN_S": os.environ.get("AFK_CHAIN_S", ""), "AFK_CHAIN_R": os.environ.get("AFK_CHAIN_R", ""), "AFK_ATTN_DECODE_SYNC": os.environ.get("AFK_ATTN_DECODE_SYNC", "")}
tot = 0.0
for name, fn in (("qkv", k_qkv), ("attn", k_attn), ("attn_two_launches", k_attn2), ("o_proj", k_o), ("gate_up", k_gu), ("down", k_d), ("lm_head", k_head)):
    for i in range(NSET): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(ITERS): fn(it % NSET)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / ITERS
    res[name] = {"us": round(us, 2), "TB/s": round(bytes_[name] / us / 1e6, 2)}
    if name not in ("attn_two_launches", "lm_head"): tot += us
res["layer_us_back_to_back"] = round(tot, 1)
res["token_ms_28_layers_plus_lm_head_0.16"] = round(tot * 28e-3 + 0.16, 3)
print(json.dumps(res))
