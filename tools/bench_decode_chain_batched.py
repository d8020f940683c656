"""per-kernel time of the launches of a BATCHED decode layer (csrc/decode_chain.hip `afk_decode_chain_*_batched`, M = 2 .. 8 sequences per step) at the AF3-7B widths,
weights rotated over NSET distinct sets so that nothing is served from the caches.   python tools/bench_decode_chain_batched.py [M] [keys]
(AFK_CHAIN_MFMA / AFK_CHAIN_MFMA_NARROW / AFK_CHAIN_S are read by the library per launch)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_flamingo_amd import _lib, ops

dev = torch.device("cuda")
BF = torch.bfloat16
H, Hq, Hkv, D, I = 3584, 28, 4, 128, 18944
nq, nk = Hq * D, Hkv * D
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
keys = int(sys.argv[2]) if len(sys.argv) > 2 else 800
NSET, ITERS = 6, int(os.environ.get("ITERS", "60"))
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=0.02: (torch.randn(*s, device=dev, generator=g) * sc).to(BF)
W = [dict(qkv=rnd(nq + 2 * nk, H), o=rnd(H, nq), gu=rnd(2 * I, H), d=rnd(H, I)) for _ in range(NSET)]
x, h = rnd(M, H, sc=1.0), rnd(M, H, sc=1.0)
x2 = torch.empty(M, H, device=dev, dtype=BF)
act, actr = torch.empty(M, I, device=dev, dtype=BF), rnd(M, I, sc=1.0)
q, o = torch.empty(M, nq, device=dev, dtype=BF), rnd(M, nq, sc=1.0)
nw, bias = rnd(H, sc=1.0), rnd(nq + 2 * nk, sc=0.1)
Smax = max(1024, keys + 64)
spad = ops.pad64(Smax)
Kc = [rnd(M, Smax, nk, sc=1.0) for _ in range(NSET)]
Vt = [rnd(M, Hkv, D, spad, sc=1.0) for _ in range(NSET)]
cos, sin = rnd(keys + 64, D, sc=1.0), rnd(keys + 64, D, sc=1.0)
pos = torch.full((M,), keys, device=dev, dtype=torch.int32)
start = torch.tensor([keys], device=dev, dtype=torch.int32)
kr = torch.tensor([[0, keys + 1]] * M, device=dev, dtype=torch.int32)
ns = int(os.environ.get("NS", "8"))
aws = torch.zeros(_lib.load().afk_attn_decode_workspace_floats(M, Hq, D, ns), device=dev, dtype=torch.float32)
st = ops._stream()
Vv = 152064
Wh = [rnd(Vv, H) for _ in range(2)]
logits = torch.empty(M, Vv, device=dev, dtype=torch.float32)

def k_norm(i): ops.rmsnorm_fwd(x, nw, 1e-6)
def k_qkv(i): _lib.call("afk_decode_chain_qkv_batched", h.data_ptr(), H, M, W[i]["qkv"].data_ptr(), H, H, bias.data_ptr(), cos.data_ptr(), sin.data_ptr(), pos.data_ptr(), q.data_ptr(), nq, Kc[i].data_ptr(), Smax * nk, Vt[i].data_ptr(), Hkv * D * spad, spad, start.data_ptr(), Hq, Hkv, D, st)
def k_attn(i): _lib.call("afk_attn_decode_fused", o.data_ptr(), nq, D, Kc[i].data_ptr(), Smax * nk, nk, D, Vt[i].data_ptr(), Hkv * D * spad, spad, q.data_ptr(), nq, D, kr.data_ptr(), M, Hq, Hkv, D, float(D ** -0.5), ns, aws.data_ptr(), st)
def k_o(i): _lib.call("afk_decode_chain_linear_residual_batched", o.data_ptr(), nq, M, W[i]["o"].data_ptr(), nq, H, nq, x.data_ptr(), H, x2.data_ptr(), H, st)
def k_gu(i): _lib.call("afk_decode_chain_gate_up_batched", h.data_ptr(), H, M, W[i]["gu"].data_ptr(), H, I, H, act.data_ptr(), I, st)
def k_d(i): _lib.call("afk_decode_chain_linear_residual_batched", actr.data_ptr(), I, M, W[i]["d"].data_ptr(), I, H, I, x.data_ptr(), H, x2.data_ptr(), H, st)
cnt = torch.zeros(1, device=dev, dtype=torch.int32)
hn = torch.empty(M, H, device=dev, dtype=BF)
def k_on(i): _lib.call("afk_decode_chain_linear_residual_norm_batched", o.data_ptr(), nq, M, W[i]["o"].data_ptr(), nq, H, nq, x.data_ptr(), H, x2.data_ptr(), H, nw.data_ptr(), 1e-6, hn.data_ptr(), H, cnt.data_ptr(), st)
def k_dn(i): _lib.call("afk_decode_chain_linear_residual_norm_batched", actr.data_ptr(), I, M, W[i]["d"].data_ptr(), I, H, I, x.data_ptr(), H, x2.data_ptr(), H, nw.data_ptr(), 1e-6, hn.data_ptr(), H, cnt.data_ptr(), st)
ssp = torch.rand(4, 8, H // 16, device=dev, dtype=torch.float32)
SS = lambda: ssp.data_ptr() if os.environ.get("SSPART", "1") == "1" else None
def k_qkvn(i): _lib.call("afk_decode_chain_qkv_norm_batched", x.data_ptr(), H, M, nw.data_ptr(), 1e-6, W[i]["qkv"].data_ptr(), H, H, bias.data_ptr(), cos.data_ptr(), sin.data_ptr(), pos.data_ptr(), q.data_ptr(), nq, Kc[i].data_ptr(), Smax * nk, Vt[i].data_ptr(), Hkv * D * spad, spad, start.data_ptr(), Hq, Hkv, D, SS(), H // 16, st)
def k_gun(i): _lib.call("afk_decode_chain_gate_up_norm_batched", x.data_ptr(), H, M, nw.data_ptr(), float(os.environ.get("EPS", "1e-6")), W[i]["gu"].data_ptr(), H, I, H, act.data_ptr(), I, SS(), H // 16, st)
def k_headn(i): _lib.call("afk_decode_chain_lm_head_norm_batched", x.data_ptr(), H, M, nw.data_ptr(), 1e-6, Wh[i % 2].data_ptr(), H, Vv, H, logits.data_ptr(), Vv, SS(), H // 16, st)
def k_os(i): _lib.call("afk_decode_chain_linear_residual_ss_batched", o.data_ptr(), nq, M, W[i]["o"].data_ptr(), nq, H, nq, x.data_ptr(), H, x2.data_ptr(), H, ssp.data_ptr(), st)
def k_ds(i): _lib.call("afk_decode_chain_linear_residual_ss_batched", actr.data_ptr(), I, M, W[i]["d"].data_ptr(), I, H, I, x.data_ptr(), H, x2.data_ptr(), H, ssp.data_ptr(), st)
def k_head(i): _lib.call("afk_decode_chain_lm_head_batched", h.data_ptr(), H, M, Wh[i % 2].data_ptr(), H, Vv, H, logits.data_ptr(), Vv, st)

bytes_ = dict(o_proj_ss=2.0 * H * nq, down_ss=2.0 * H * I, qkv_normpro=2.0 * (nq + 2 * nk) * H, gate_up_normpro=2.0 * 2 * I * H, lm_head_normpro=2.0 * Vv * H, o_proj_norm=2.0 * H * nq, down_norm=2.0 * H * I, norm=4.0 * M * H, qkv=2.0 * (nq + 2 * nk) * H, attn=2.0 * 2 * (keys + 1) * nk * M, o_proj=2.0 * H * nq, gate_up=2.0 * 2 * I * H, down=2.0 * H * I, lm_head=2.0 * Vv * H)
res = {"M": M, "keys": keys, "nsplit": ns, **{k: os.environ.get(k, "") for k in ("AFK_CHAIN_MFMA", "AFK_CHAIN_MFMA_NARROW", "AFK_CHAIN_S", "AFK_CHAIN_FORM")}}
only = os.environ.get("ONLY", "").split(",") if os.environ.get("ONLY") else None
tot = 0.0
for name, fn in (("norm", k_norm), ("qkv", k_qkv), ("attn", k_attn), ("o_proj", k_o), ("gate_up", k_gu), ("down", k_d), ("o_proj_norm", k_on), ("down_norm", k_dn), ("o_proj_ss", k_os), ("down_ss", k_ds), ("qkv_normpro", k_qkvn), ("gate_up_normpro", k_gun), ("lm_head", k_head), ("lm_head_normpro", k_headn)):
    if only and name not in only: continue
    for i in range(NSET): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(ITERS): fn(it % NSET)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / ITERS
    res[name] = {"us": round(us, 2), "TB/s": round(bytes_[name] / us / 1e6, 2)}
    if name not in ("lm_head", "o_proj_norm", "down_norm", "qkv_normpro", "gate_up_normpro", "lm_head_normpro", "o_proj_ss", "down_ss"): tot += us * (2 if name == "norm" else 1)
res["layer_us_back_to_back"] = round(tot, 1)
print(json.dumps(res))
