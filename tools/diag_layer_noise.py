#!/usr/bin/env python
"""Teacher-forced rounding noise of every sub-stage of ONE decoder layer: each stage gets OUR bf16 output of the previous stage as input; its
output is compared with the same stage in fp32 torch, beside the stage as eager bf16 torch (what the reference model runs).  Diagnostic only."""
import os, sys, math
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
BF = torch.bfloat16

def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / b.norm())

def main(B=4):
    from audio_flamingo_amd import ops
    from audio_flamingo_amd.frontend import LogMelFrontend
    from audio_flamingo_amd.modeling import AudioFlamingo3ForConditionalGeneration as Mine
    dev = torch.device("cuda", 0)
    cfg = bench.af3_7b_config(1, 1)
    m = Mine(cfg, device=dev, init_seed=3)
    g = torch.Generator(device=dev).manual_seed(4)
    with torch.no_grad():
        for blk in m.arena.order:
            if blk.key.endswith(".bias"):
                blk.data.copy_((0.02 * torch.randn(blk.shape, device=dev, generator=g)).to(BF))
            elif blk.key.endswith("norm.weight"):
                blk.data.copy_((1 + 0.05 * torch.randn(blk.shape, device=dev, generator=g)).to(BF))
    m.arena.step_counter += 1
    waves, ids, labels = bench.synthetic_batch(B, 0, dev, 1)
    S = ids.shape[1]
    with torch.no_grad():
        out = m(input_ids=ids, input_features=LogMelFrontend(dev)(waves, out_dtype=BF), output_hidden_states=True)
    x = out.hidden_states[0].reshape(B * S, -1).contiguous()
    A = lambda k: m.arena["model.language_model.layers.0." + k].data
    Hq, Hkv, D, eps = 28, 4, 128, 1e-6
    cos, sin = m._rope_tables(S)
    rows = []
    def rms(xx, w, dt):
        x32 = xx.float() if dt == torch.float32 else xx
        v = x32.float()
        y = v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + eps)
        return w.to(dt) * y.to(dt)
    # 1 rmsnorm
    h, _ = ops.rmsnorm_fwd(x, A("input_layernorm.weight"), eps)
    rows.append(("rmsnorm1", rel(h, rms(x, A("input_layernorm.weight"), torch.float32)), rel(rms(x, A("input_layernorm.weight"), BF), rms(x, A("input_layernorm.weight"), torch.float32))))
    # 2 qkv
    Wqkv, bqkv = A("self_attn.qkv.weight"), A("self_attn.qkv.bias")
    qkv = ops.gemm_nt(h, Wqkv, bias=bqkv)
    ref32 = h.float() @ Wqkv.float().t() + bqkv.float()
    rows.append(("qkv gemm+bias", rel(qkv, ref32), rel(torch.nn.functional.linear(h, Wqkv, bqkv), ref32)))
    # 3 rope
    qkv_in = qkv.clone()
    ops.rope_(qkv, cos, sin, S=S, nheads=Hq + Hkv, D=D, pos=None)
    def rope_t(t, dt, c, s):   # t [B,S,H,D]
        c, s = c.to(dt)[None, :, None, :], s.to(dt)[None, :, None, :]
        t = t.to(dt)
        t1, t2 = t[..., : D // 2], t[..., D // 2:]
        return t * c + torch.cat((-t2, t1), -1) * s
    # fp32 truth uses UNROUNDED cos/sin (the fp32 reference keeps them in fp32, modeling_qwen2.py:91-102)
    inv = 1.0 / (1e6 ** (torch.arange(0, D, 2, device=dev, dtype=torch.float32) / D))
    fr = torch.arange(S, device=dev, dtype=torch.float32)[:, None] * inv[None]
    emb = torch.cat((fr, fr), -1)
    c32, s32 = emb.cos(), emb.sin()
    print("cos table shape", tuple(cos.shape), cos.dtype, "max |cos - bf16(cos32)|", float((cos.float().reshape(S, -1)[:, :D] - c32.to(BF).float()).abs().max()) if cos.numel() >= S * D else "n/a")
    nqk = (Hq + Hkv) * D
    t_in = qkv_in[:, :nqk].reshape(B, S, Hq + Hkv, D)
    r32 = rope_t(t_in, torch.float32, c32, s32)
    r32_tab = rope_t(t_in, torch.float32, c32.to(BF).float(), s32.to(BF).float())
    r16 = rope_t(t_in, BF, c32.to(BF), s32.to(BF))
    got = qkv[:, :nqk].reshape(B, S, Hq + Hkv, D)
    rows.append(("rope (vs fp32 tables)", rel(got, r32), rel(r16, r32)))
    rows.append(("rope (vs bf16-rounded tables in fp32)", rel(got, r32_tab), rel(r16, r32_tab)))
    # 4 attention
    o, lse = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=True, kv_len=None, kv_lo=None)
    q = qkv[:, : Hq * D].reshape(B, S, Hq, D).transpose(1, 2)
    k = qkv[:, Hq * D: nqk].reshape(B, S, Hkv, D).transpose(1, 2)
    v = qkv[:, nqk:].reshape(B, S, Hkv, D).transpose(1, 2)
    o32 = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float(), is_causal=True, enable_gqa=True).transpose(1, 2).reshape(B * S, -1)
    o16 = torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=True).transpose(1, 2).reshape(B * S, -1)
    rows.append(("attention", rel(o, o32), rel(o16, o32)))
    # 5 o_proj + residual
    Wo = A("self_attn.o_proj.weight")
    x2 = ops.gemm_nt(o, Wo, residual=x)
    ref32 = o.float() @ Wo.float().t() + x.float()
    rows.append(("o_proj + residual", rel(x2, ref32), rel(x + torch.nn.functional.linear(o, Wo), ref32)))
    rows.append(("o_proj alone (ours - x)", rel(x2.float() - x.float(), o.float() @ Wo.float().t()), rel(torch.nn.functional.linear(o, Wo), o.float() @ Wo.float().t())))
    # 6 rmsnorm2
    h2, _ = ops.rmsnorm_fwd(x2, A("post_attention_layernorm.weight"), eps)
    rows.append(("rmsnorm2", rel(h2, rms(x2, A("post_attention_layernorm.weight"), torch.float32)), rel(rms(x2, A("post_attention_layernorm.weight"), BF), rms(x2, A("post_attention_layernorm.weight"), torch.float32))))
    # 7 gate|up + swiglu
    wgu = A("mlp.gate_up.weight")
    a = torch.empty((h2.shape[0], wgu.shape[0] // 2), device=dev, dtype=BF)
    gu = ops.gemm_nt(h2, wgu, swiglu_fwd_out=a)
    gu32 = h2.float() @ wgu.float().t()
    rows.append(("gate|up gemm", rel(gu, gu32), rel(torch.nn.functional.linear(h2, wgu), gu32)))
    a_sep = ops.silu_mul_fwd(gu)
    print("swiglu layout check: fused == separate", bool(torch.equal(a, a_sep)))
    I = wgu.shape[0] // 2
    # which half is gate?  take the layout the separate kernel implements: compare both interpretations
    cand = {"gate|up halves": torch.nn.functional.silu(gu32[:, :I]) * gu32[:, I:], "interleaved": torch.nn.functional.silu(gu32[:, 0::2]) * gu32[:, 1::2]}
    best = min(cand, key=lambda kk: rel(a, cand[kk]))
    a32 = cand[best]
    g16 = torch.nn.functional.linear(h2, wgu)
    a16 = (torch.nn.functional.silu(g16[:, :I]) * g16[:, I:]) if best == "gate|up halves" else (torch.nn.functional.silu(g16[:, 0::2]) * g16[:, 1::2])
    rows.append((f"swiglu out ({best})", rel(a, a32), rel(a16, a32)))
    # 8 down + residual
    Wd = A("mlp.down_proj.weight")
    x3 = ops.gemm_nt(a, Wd, residual=x2)
    ref32 = a.float() @ Wd.float().t() + x2.float()
    rows.append(("down_proj + residual", rel(x3, ref32), rel(x2 + torch.nn.functional.linear(a, Wd), ref32)))
    rows.append(("down_proj alone", rel(x3.float() - x2.float(), a.float() @ Wd.float().t()), rel(torch.nn.functional.linear(a, Wd), a.float() @ Wd.float().t())))
    print(f"{'stage (teacher-forced)':44s} {'ours vs fp32':>14s} {'torch bf16 vs fp32':>20s}")
    for n, a_, b_ in rows:
        print(f"{n:44s} {a_:14.6f} {b_:20.6f}")
    print("norms: x", float(x.float().norm()), "x2", float(x2.float().norm()), "x3", float(x3.float().norm()), "text-row rms", float(x.float().reshape(B, S, -1)[:, :9].pow(2).mean().sqrt()),
          "audio-row rms", float(x.float().reshape(B, S, -1)[:, 9:759].pow(2).mean().sqrt()))

if __name__ == "__main__":
    main()
