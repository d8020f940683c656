"""ORACLE — test infrastructure only (never imported by the product path).

CPU fp32 restatement of the Flamingo conditioning blocks of BASELINE config 4 (AF1/AF2: Perceiver resampler + tanh-gated
cross-attention).  **PARITY UNPINNED with respect to AF1/AF2**: their source lives on un-mounted branches of the reference and
no port exists in the installed wheels.  What this file restates, line by line, is the structural
stand-in that is available - HuggingFace's Flamingo re-implementation in transformers 5.15:
    P:<line>  = transformers/models/idefics/perceiver.py
    M:<line>  = transformers/models/idefics/modeling_idefics.py
and tests/test_oracle_cpu.py pins it to those live modules.  It is the declared spec of the HIP blocks in
audio_flamingo_amd/flamingo.py, nothing more.
"""
import torch
import torch.nn.functional as F


def perceiver_attention(sd, pfx, context, latents, n_heads, head_dim):
    """P:128-168 (qk_layer_norms False)"""
    E = context.shape[-1]
    c = F.layer_norm(context, (E,), sd[pfx + "context_layer_norm.weight"], sd[pfx + "context_layer_norm.bias"])   # P:143
    l = F.layer_norm(latents, (E,), sd[pfx + "latents_layer_norm.weight"], sd[pfx + "latents_layer_norm.bias"])   # P:144
    B = c.shape[0]
    q = F.linear(l, sd[pfx + "q_proj.weight"])                                                                     # P:149
    kv = torch.cat([c, l], dim=-2)                                                                                 # P:150-151
    k, v = F.linear(kv, sd[pfx + "k_proj.weight"]), F.linear(kv, sd[pfx + "v_proj.weight"])
    q, k, v = [x.reshape(B, x.shape[1], n_heads, head_dim).transpose(1, 2) for x in (q, k, v)]                    # P:156
    s = torch.einsum("...id,...jd->...ij", q * head_dim ** -0.5, k)                                               # P:162
    a = (s - s.amax(-1, keepdim=True)).softmax(-1)                                                                 # P:163-164
    o = torch.einsum("...ij,...jd->...id", a, v).transpose(1, 2).flatten(-2)                                       # P:167-168
    return F.linear(o, sd[pfx + "output_proj.weight"])


def perceiver_resampler(sd, context, n_heads, head_dim, pfx=""):
    """P:93-102"""
    B, _, E = context.shape
    lat = sd[pfx + "latents"].repeat(B, 1, 1)                                                                      # P:95
    i = 0
    while f"{pfx}blocks.{i}.0.q_proj.weight" in sd:
        lat = perceiver_attention(sd, f"{pfx}blocks.{i}.0.", context, lat, n_heads, head_dim) + lat                # P:99
        p = f"{pfx}blocks.{i}.1."
        h = F.layer_norm(lat, (E,), sd[p + "ln.weight"], sd[p + "ln.bias"])                                        # P:182
        lat = F.linear(F.relu(F.linear(h, sd[p + "fc.weight"])), sd[p + "c_proj.weight"]) + lat                    # P:183-185, 100
        i += 1
    return F.layer_norm(lat, (E,), sd[pfx + "layer_norm.weight"], sd[pfx + "layer_norm.bias"])                     # P:102


def rms_norm(x, w, eps):
    v = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(v + eps))


def gated_cross_attention(sd, x, media, keep_mask, gate, n_heads, eps, pfx=""):
    """M:776-802.  keep_mask bool [B, S, Sk] (True = may attend), gate [B, S] (0 = token attends to no media)"""
    B, S, H = x.shape
    D = H // n_heads
    r = x
    h = rms_norm(x, sd[pfx + "input_layernorm.weight"], eps)                                                        # M:781
    q = F.linear(h, sd[pfx + "cross_attn.q_proj.weight"]).view(B, S, n_heads, D).transpose(1, 2)                    # M:575
    k = F.linear(media, sd[pfx + "cross_attn.k_proj.weight"]).view(B, -1, n_heads, D).transpose(1, 2)               # M:581
    v = F.linear(media, sd[pfx + "cross_attn.v_proj.weight"]).view(B, -1, n_heads, D).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * D ** -0.5                                                                       # no rotary for cross attention, M:590
    s = s.masked_fill(~keep_mask[:, None], float("-inf"))
    p = torch.softmax(s, -1)
    p = torch.nan_to_num(p, nan=0.0)  # rows with no visible media; zeroed by the gate below anyway (M:792)
    o = (p @ v).transpose(1, 2).reshape(B, S, H)
    a = F.linear(o, sd[pfx + "cross_attn.o_proj.weight"])
    a = a.masked_fill((gate == 0)[:, :, None], 0.0)                                                                 # M:792
    x = r + torch.tanh(sd[pfx + "alpha_cross_attn"]) * a                                                            # M:793
    r = x
    h = rms_norm(x, sd[pfx + "post_attention_layernorm.weight"], eps)                                               # M:797
    m = F.linear(F.silu(F.linear(h, sd[pfx + "mlp.gate_proj.weight"])) * F.linear(h, sd[pfx + "mlp.up_proj.weight"]),
                 sd[pfx + "mlp.down_proj.weight"])                                                                  # M:798
    return r + torch.tanh(sd[pfx + "alpha_dense"]) * m                                                              # M:800


def qwen2_layer(sd, pfx, x, n_heads, n_kv, head_dim, eps, theta):
    """one Qwen2DecoderLayer on a FUSED q|k|v projection (the layout audio_flamingo_amd/flamingo_icl.py stores): Q2 = modeling_qwen2.py:258-298"""
    from oracle.af3_oracle import _sdpa, rotate_half

    B, S, H = x.shape
    nq, nkv = n_heads * head_dim, n_kv * head_dim
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = torch.arange(S, dtype=torch.float32)[:, None] * inv[None]
    emb = torch.cat((fr, fr), -1)
    cos, sin = emb.cos()[None, None], emb.sin()[None, None]
    r = x
    h = sd[pfx + "input_layernorm.weight"] * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))
    qkv = F.linear(h, sd[pfx + "qkv.weight"], sd[pfx + "qkv.bias"])
    q = qkv[..., :nq].view(B, S, n_heads, head_dim).transpose(1, 2)
    k = qkv[..., nq: nq + nkv].view(B, S, n_kv, head_dim).transpose(1, 2)
    v = qkv[..., nq + nkv:].view(B, S, n_kv, head_dim).transpose(1, 2)
    q, k = q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin
    g = n_heads // n_kv
    keep = torch.tril(torch.ones(S, S, dtype=torch.bool))[None, None]
    o = _sdpa(q, k.repeat_interleave(g, 1), v.repeat_interleave(g, 1), keep, head_dim ** -0.5).transpose(1, 2).reshape(B, S, nq)
    x = r + F.linear(o, sd[pfx + "o_proj.weight"])
    r = x
    h = sd[pfx + "post_attention_layernorm.weight"] * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))
    gu = F.linear(h, sd[pfx + "gate_up.weight"])
    I = gu.shape[-1] // 2
    return r + F.linear(F.silu(gu[..., :I]) * gu[..., I:], sd[pfx + "down_proj.weight"])


def icl_forward(sd, c, input_ids, audio_features, labels):
    """the assembled config-4 step of audio_flamingo_amd/flamingo_icl.py: projection -> resampler -> decoder with gated cross-attention
    blocks -> lm_head -> shifted CE.  (Builder-declared wiring; the blocks themselves are pinned to the Idefics stand-in above.)"""
    B, S = input_ids.shape
    nc, T, De = audio_features.shape[1:]
    H, L = c["hidden"], c["n_latents"]
    f = F.linear(audio_features.reshape(B * nc, T, De), sd["audio_proj.weight"])
    media = perceiver_resampler(sd, f, c["resampler_heads"], c["resampler_head_dim"], pfx="resampler.").reshape(B, nc * L, H)
    ci = (input_ids == c["audio_marker_id"]).long().cumsum(-1) - 1
    gate = (ci >= 0).long()
    keys = torch.arange(nc * L)[None, None, :]
    keep = (keys >= (ci.clamp_min(0) * L)[..., None]) & (keys < ((ci.clamp_min(0) + 1) * L)[..., None]) & (ci >= 0)[..., None]
    x = sd["embed_tokens.weight"][input_ids]
    for i in range(c["layers"]):
        if i % c["xattn_every"] == 0:
            x = gated_cross_attention(sd, x, media, keep, gate, c["xattn_heads"], c["rms_eps"], pfx=f"xattn.{i}.")
        x = qwen2_layer(sd, f"layers.{i}.", x, c["heads"], c["kv_heads"], c["head_dim"], c["rms_eps"], c["rope_theta"])
    x = sd["norm.weight"] * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + c["rms_eps"]))
    logits = F.linear(x, sd["lm_head.weight"])
    lab = F.pad(labels, (0, 1), value=-100)[..., 1:].reshape(-1)
    return F.cross_entropy(logits.view(-1, logits.shape[-1]), lab, ignore_index=-100), logits
