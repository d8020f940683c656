"""AFK_EXACT_FP32=1 - the AF3 inference forward in exact fp32 on the `afk_x32_*` kernels (csrc/exact_f32.hip).

SURVEY.md §8c: "An fp32 mode of our kernels (f32 MFMA v_mfma_f32_32x32x2_f32, exact fp32) should give bit-exact tokens unconditionally on the tiny
config."  The bf16 product path matches the reference's token ids only where the reference's own top-1 / top-2 gap exceeds bf16 noise ("confident"
positions); this mode removes every bf16 rounding point between the log-mel features and the logits - activations fp32, the checkpoint's bf16 weight
VALUES widened in registers, all sums fp32 - so its argmax equals the fp32 reference's at EVERY valid position and `generate()` reproduces the reference's
greedy ids with no filter (tests/test_exact_gpu.py).  A verification mode: one wave per output tile / query row, no LDS tiling, no KV cache (greedy decoding
recomputes the prefix, as GenerationMixin does with use_cache=False).  Inference only - there is no backward.

Same algorithm, line for line, as the bf16 path (modeling.py / functional.py), i.e. as the reference:
  conv stem + GELU + position table  modeling_audioflamingo3.py:380-385     encoder layer (pre-LN, q scaled by d^-1/2, k_proj without bias)  :117-245
  avg-pool + LayerNorm :401-403      projector :435-439                     valid rows + placeholder scatter :483-545
  Qwen2 decoder layer (RMSNorm, RoPE rotate-half with fp32 tables, causal GQA attention with fp32 softmax, SwiGLU)  modeling_qwen2.py:46-48,91-135,195-298
  final norm + lm_head :398, modeling_audioflamingo3.py:625-627
Integer index plumbing (window lengths, placeholder ranks, padding intervals) is torch on the device, as in the bf16 path.
"""
from __future__ import annotations

import os

import torch

from . import _lib, ops
from ._lib import AfkError

ENABLED = os.environ.get("AFK_EXACT_FP32", "0") == "1"
F32 = torch.float32


def _st():
    return ops._stream()


def linear(x, w, bias=None, *, residual=None, res_mod=0, alpha=1.0, gelu=False):
    """x fp32 [M, K] (row stride may exceed K), w bf16 [N, K] -> fp32 [M, N]"""
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty((M, N), device=x.device, dtype=F32)
    _lib.call("afk_x32_linear", x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), N, M, N, K, 0 if bias is None else bias.data_ptr(),
              0 if residual is None else residual.data_ptr(), 0 if residual is None else residual.stride(0), int(res_mod), float(alpha), int(gelu), _st())
    return out


def norm(x, w, b, eps, rms):
    rows, D = x.shape
    y = torch.empty_like(x)
    _lib.call("afk_x32_norm", x.data_ptr(), w.data_ptr(), 0 if b is None else b.data_ptr(), y.data_ptr(), rows, D, float(eps), int(rms), _st())
    return y


def attention(qkv, B, S, Hq, Hkv, D, scale, causal, kv_lo=None, kv_len=None):
    """qkv fp32 [B * S, (Hq + 2 Hkv) D] fused rows -> fp32 [B * S, Hq D]"""
    ld = qkv.stride(0)
    o = torch.empty((B * S, Hq * D), device=qkv.device, dtype=F32)
    k = qkv[:, Hq * D:]
    v = qkv[:, (Hq + Hkv) * D:]
    _lib.call("afk_x32_attention", qkv.data_ptr(), ld, k.data_ptr(), ld, v.data_ptr(), ld, o.data_ptr(), Hq * D, B, S, Hq, Hkv, D, float(scale), int(causal),
              0 if kv_lo is None else kv_lo.data_ptr(), 0 if kv_len is None else kv_len.data_ptr(), _st())
    return o


@torch.no_grad()
def logits(model, input_ids, input_features=None, input_features_mask=None, attention_mask=None) -> torch.Tensor:
    """-> fp32 logits [B, S, V] of the AF3 forward (modeling_audioflamingo3.py:584-642 without labels)"""
    if type(model).__name__ != "AudioFlamingo3ForConditionalGeneration":
        raise AfkError("AFK_EXACT_FP32: the exact fp32 mode covers the AF3 forward only")
    model._require_hip()
    dev, a = model.device_, model.arena
    at, pj, lm = model._at, model._pj, model._lm
    A = lambda k: a[k].data
    ids = input_ids.to(dev)
    B, S = ids.shape
    ids_flat = ids.reshape(-1).contiguous()
    audio, src = None, None
    if input_features is not None:
        feats = input_features.to(dev, F32).contiguous()
        W, C, T = feats.shape
        E, He, De = model.E, model.enc_heads, model.E // model.enc_heads
        T2 = (T - 1) // 2 + 1
        if T2 != model.max_pos:
            raise AfkError(f"input_features: {T} frames give {T2} encoder positions, the position table holds {model.max_pos}")
        kv_len_w = n_tok = None
        if input_features_mask is not None:
            L1 = (input_features_mask.to(dev).sum(-1) - 1) // 2 + 1                     # :374-377
            kv_len_w = L1.to(torch.int32).contiguous()
            n_tok = (L1 - 2) // 2 + 1                                                   # :410-416
        x1 = torch.empty((W * T, E), device=dev, dtype=F32)
        _lib.call("afk_x32_conv3_gelu", feats.data_ptr(), 1, A(at + "conv1.weight").data_ptr(), A(at + "conv1.bias").data_ptr(), 0, x1.data_ptr(), W, C, T, E, 1, _st())
        x = torch.empty((W * T2, E), device=dev, dtype=F32)
        _lib.call("afk_x32_conv3_gelu", x1.data_ptr(), 0, A(at + "conv2.weight").data_ptr(), A(at + "conv2.bias").data_ptr(), model.embed_positions.data_ptr(),
                  x.data_ptr(), W, E, T, E, 2, _st())
        del x1
        for i in range(model.enc_layers):
            p = f"{at}layers.{i}."
            h = norm(x, A(p + "self_attn_layer_norm.weight"), A(p + "self_attn_layer_norm.bias"), 1e-5, False)
            qkv = linear(h, A(p + "self_attn.qkv.weight"), A(p + "self_attn.qkv.bias"))   # the k third of the fused bias is zero (k_proj has no bias, :112)
            # the reference scales q before the head reshape and calls the backend with scaling 1.0 (:142,181): (q * s) . k == (q . k) * s up to one fp32 rounding
            o = attention(qkv, W, T2, He, He, De, De ** -0.5, False, None, kv_len_w)
            x = linear(o, A(p + "self_attn.out_proj.weight"), A(p + "self_attn.out_proj.bias"), residual=x)
            h = norm(x, A(p + "final_layer_norm.weight"), A(p + "final_layer_norm.bias"), 1e-5, False)
            f = linear(h, A(p + "fc1.weight"), A(p + "fc1.bias"), gelu=True)
            x = linear(f, A(p + "fc2.weight"), A(p + "fc2.bias"), residual=x)
        T3 = T2 // 2
        pooled = torch.empty((W * T3, E), device=dev, dtype=F32)
        _lib.call("afk_x32_avgpool2", x.data_ptr(), pooled.data_ptr(), W, T2, E, _st())
        h = norm(pooled, A(at + "layer_norm.weight"), A(at + "layer_norm.bias"), 1e-5, False)
        h = linear(h, A(pj + "linear_1.weight"), A(pj + "linear_1.bias"), gelu=True)
        audio = linear(h, A(pj + "linear_2.weight"), A(pj + "linear_2.bias"))              # [W * T3, H], padded windows included
        src, cnt = ops.placeholder_scan(ids_flat, model.audio_token_id)
        if n_tok is not None:   # rank r among the placeholders -> row (window, t) of the padded buffer (:483-486), as modeling.forward does
            csum = torch.cumsum(n_tok, 0)
            r = src.clamp_min(0).to(torch.int64)
            win = torch.searchsorted(csum, r, right=True).clamp_max(n_tok.numel() - 1)
            row = win * T3 + (r - (csum - n_tok)[win])
            src = torch.where(src >= 0, row.to(torch.int32), src).contiguous()
            expected = int(csum[-1])
        else:
            expected = int(audio.shape[0])
        if int(cnt.item()) != expected:
            raise ValueError(f"Audio features and audio tokens do not match, tokens: {int(cnt.item())}, features: {expected}")
    H, Hq, Hkv, D, I = model.H, model.Hq, model.Hkv, model.D, model.I
    x = torch.empty((B * S, H), device=dev, dtype=F32)
    _lib.call("afk_x32_embed_scatter", ids_flat.data_ptr(), 0 if src is None else src.data_ptr(), 0 if audio is None else audio.data_ptr(),
              A(lm + "embed_tokens.weight").data_ptr(), x.data_ptr(), B * S, H, _st())
    kv_lo = kv_len = None
    if attention_mask is not None:
        iv = model._mask_intervals(attention_mask)
        if iv is not None:
            kv_lo, kv_len = (t.to(dev, torch.int32).contiguous() for t in iv)
    # RoPE tables in fp32, NOT rounded to bf16: the fp32 reference keeps cos / sin in the activation dtype (modeling_qwen2.py:102)
    inv = 1.0 / (model.rope_theta ** (torch.arange(0, D, 2, device=dev, dtype=F32) / D))
    fr = torch.arange(S, device=dev, dtype=F32)[:, None] * inv[None, :]
    emb = torch.cat([fr, fr], dim=-1)
    cos, sin = emb.cos().contiguous(), emb.sin().contiguous()
    for i in range(model.dec_layers):
        p = f"{lm}layers.{i}."
        h = norm(x, A(p + "input_layernorm.weight"), None, model.rms_eps, True)
        qkv = linear(h, A(p + "self_attn.qkv.weight"), A(p + "self_attn.qkv.bias"))
        _lib.call("afk_x32_rope", qkv.data_ptr(), qkv.stride(0), cos.data_ptr(), sin.data_ptr(), B * S, S, Hq + Hkv, D, _st())
        o = attention(qkv, B, S, Hq, Hkv, D, D ** -0.5, True, kv_lo, kv_len)
        x = linear(o, A(p + "self_attn.o_proj.weight"), residual=x)
        h = norm(x, A(p + "post_attention_layernorm.weight"), None, model.rms_eps, True)
        gu = linear(h, A(p + "mlp.gate_up.weight"))
        act = torch.empty((B * S, I), device=dev, dtype=F32)
        _lib.call("afk_x32_silu_mul", gu.data_ptr(), act.data_ptr(), B * S, I, _st())
        x = linear(act, A(p + "mlp.down_proj.weight"), residual=x)
    x = norm(x, A(lm + "norm.weight"), None, model.rms_eps, True)
    return linear(x, A("lm_head.weight")).reshape(B, S, model.V)


@torch.no_grad()
def greedy_generate(model, input_ids, input_features=None, input_features_mask=None, attention_mask=None, max_new_tokens=20, eos_token_id=None,
                    pad_token_id=None):
    """greedy search without a cache (GenerationMixin with use_cache=False): re-run the prefix in exact fp32, append the argmax of the last position.
    GenerationMixin semantics for finished rows (generation/utils.py:2730-2830): eos_token_id may be an int or a list; a row is finished once it has
    emitted ANY of them and emits pad_token_id from then on (the first EOS when no pad id is configured, as modeling.generate does)."""
    dev = model.device_
    ids = input_ids.to(dev)
    att = None if attention_mask is None else attention_mask.to(dev)
    eos = None if eos_token_id is None else torch.as_tensor(eos_token_id, device=dev, dtype=ids.dtype).reshape(-1)
    pad = int(pad_token_id) if pad_token_id is not None else (int(eos[0]) if eos is not None else None)
    done = torch.zeros(ids.shape[0], dtype=torch.bool, device=dev)
    for _ in range(int(max_new_tokens)):
        lg = logits(model, ids, input_features, input_features_mask, att)
        nxt = lg[:, -1].argmax(-1)
        if eos is not None:
            nxt = torch.where(done, torch.full_like(nxt, pad), nxt)
            done |= torch.isin(nxt, eos) & ~done
        ids = torch.cat([ids, nxt[:, None]], 1)
        if att is not None:
            att = torch.cat([att, torch.ones_like(att[:, :1])], 1)
        if eos is not None and bool(done.all()):
            break
    return ids
