"""AF1/AF2-style conditioning blocks on the MI355X kernels (BASELINE config 4: "gated-xattn + Perceiver resampler, 4 few-shot
clips per sample").

PARITY STATUS: **unpinned**.  The AF1/AF2 source (OpenFlamingo forks on branches that are not mounted) is not
available and transformers ships no AF1/AF2 port.  These modules therefore follow the nearest structural stand-in that IS
available, HuggingFace's Flamingo re-implementation (Idefics), parameter names included:
    IdeficsPerceiverResampler / IdeficsPerceiverAttention / IdeficsMLP      transformers/models/idefics/perceiver.py:46-187
    IdeficsGatedCrossAttentionLayer (+ IdeficsAttention cross form, MLP)    transformers/models/idefics/modeling_idefics.py:474-616,678-802
and are parity-checked against that stand-in (oracle/flamingo_oracle.py), nothing more.  Arithmetic = afk kernels through
autograd_ops (MFMA GEMMs incl. the NN/TN backward forms, cross-attention with per-query key ranges, LN/RMSNorm, ReLU, SwiGLU,
tanh-gated residual).
"""
from __future__ import annotations

import torch
from torch import nn

from . import autograd_ops as A

BF = torch.bfloat16


def _p(*shape, std=0.02, device=None):
    return nn.Parameter((torch.randn(*shape, device=device) * std).to(BF))


def _ones(n, device):
    return nn.Parameter(torch.ones(n, device=device, dtype=BF))


def _zeros(n, device):
    return nn.Parameter(torch.zeros(n, device=device, dtype=BF))


class _LN(nn.Module):
    def __init__(self, d, device):
        super().__init__()
        self.weight, self.bias = _ones(d, device), _zeros(d, device)

    def forward(self, x):
        return A.layer_norm(x, self.weight, self.bias, 1e-5)


class _Lin(nn.Module):
    def __init__(self, i, o, device):
        super().__init__()
        self.weight = _p(o, i, device=device)

    def forward(self, x, residual=None):
        return A.linear(x, self.weight, None, residual)


class PerceiverAttention(nn.Module):
    """perceiver.py:106-168 (qk_layer_norms off): latents attend to cat(context, latents)"""

    def __init__(self, embed_dim, n_heads, head_dim, device):
        super().__init__()
        self.n_heads, self.head_dim = n_heads, head_dim
        self.context_layer_norm, self.latents_layer_norm = _LN(embed_dim, device), _LN(embed_dim, device)
        inner = n_heads * head_dim
        self.q_proj, self.k_proj, self.v_proj = _Lin(embed_dim, inner, device), _Lin(embed_dim, inner, device), _Lin(embed_dim, inner, device)
        self.output_proj = _Lin(inner, embed_dim, device)

    def forward(self, context, latents, B, T, L, residual):
        """context [B*T, E], latents [B*L, E] -> [B*L, E] (+ residual)"""
        E = context.shape[1]
        c = self.context_layer_norm(context)
        l = self.latents_layer_norm(latents)
        kv_in = torch.cat([c.view(B, T, E), l.view(B, L, E)], dim=1).reshape(B * (T + L), E)  # layout plumbing (perceiver.py:150)
        q = self.q_proj(l)
        k = self.k_proj(kv_in)
        v = self.v_proj(kv_in)
        o = A.cross_attention(q, k, v, B=B, Sq=L, Sk=T + L, H=self.n_heads, D=self.head_dim, scale=self.head_dim ** -0.5)
        return self.output_proj(o, residual=residual)


class PerceiverMLP(nn.Module):
    """perceiver.py:171-187: LN -> fc -> ReLU -> c_proj (no biases)"""

    def __init__(self, embed_dim, intermediate, device):
        super().__init__()
        self.ln = _LN(embed_dim, device)
        self.fc, self.c_proj = _Lin(embed_dim, intermediate, device), _Lin(intermediate, embed_dim, device)

    def forward(self, x, residual):
        return self.c_proj(A.relu(self.fc(self.ln(x))), residual=residual)


class PerceiverResampler(nn.Module):
    """perceiver.py:46-102: [B, T, E] context -> [B, n_latents, E]"""

    def __init__(self, embed_dim, depth, n_heads, head_dim, n_latents, device="cuda"):
        super().__init__()
        self.embed_dim, self.n_latents = embed_dim, n_latents
        self.latents = nn.Parameter(torch.randn(n_latents, embed_dim, device=device).to(BF))
        self.blocks = nn.ModuleList([nn.ModuleList([PerceiverAttention(embed_dim, n_heads, head_dim, device),
                                                    PerceiverMLP(embed_dim, embed_dim * 4, device)]) for _ in range(depth)])
        self.layer_norm = _LN(embed_dim, device)

    def forward(self, context):
        B, T, E = context.shape
        ctx2 = context.reshape(B * T, E).contiguous()
        lat = self.latents.unsqueeze(0).expand(B, -1, -1).reshape(B * self.n_latents, E).contiguous()  # perceiver.py:95 (.repeat)
        for attn, ff in self.blocks:
            lat = attn(ctx2, lat, B, T, self.n_latents, residual=lat)
            lat = ff(lat, residual=lat)
        return self.layer_norm(lat).view(B, self.n_latents, E)


class _RMS(nn.Module):
    def __init__(self, d, eps, device):
        super().__init__()
        self.weight, self.eps = _ones(d, device), eps

    def forward(self, x):
        return A.rms_norm(x, self.weight, self.eps)


class _CrossAttn(nn.Module):
    def __init__(self, hidden, heads, device):
        super().__init__()
        self.heads, self.head_dim = heads, hidden // heads
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = (_Lin(hidden, hidden, device) for _ in range(4))


class _GLU(nn.Module):
    def __init__(self, hidden, inter, device):
        super().__init__()
        self.gate_proj, self.up_proj, self.down_proj = _Lin(hidden, inter, device), _Lin(hidden, inter, device), _Lin(inter, hidden, device)


class GatedCrossAttentionBlock(nn.Module):
    """modeling_idefics.py:678-802: x += tanh(a_x) * no_media_zero(XAttn(norm(x), media)); x += tanh(a_ff) * MLP(norm(x))"""

    def __init__(self, hidden, heads, intermediate, eps=1e-6, alpha_type="vector", device="cuda"):
        super().__init__()
        self.hidden = hidden
        self.cross_attn = _CrossAttn(hidden, heads, device)
        self.mlp = _GLU(hidden, intermediate, device)
        self.input_layernorm, self.post_attention_layernorm = _RMS(hidden, eps, device), _RMS(hidden, eps, device)
        n = hidden if alpha_type == "vector" else 1
        self.alpha_cross_attn = nn.Parameter(torch.zeros((1, 1, n) if alpha_type == "vector" else (1,), device=device, dtype=BF))
        self.alpha_dense = nn.Parameter(torch.zeros((1, 1, n) if alpha_type == "vector" else (1,), device=device, dtype=BF))

    def forward(self, hidden_states, media, key_range, gate):
        """hidden_states [B, S, H]; media [B, Sk, H]; key_range int32 [B, S, 2] = keys each text token may see (its own clip's
        latents; empty for tokens before any clip); gate int32 [B, S] = 0 for tokens attending to no media (:792)"""
        B, S, H = hidden_states.shape
        Sk = media.shape[1]
        x = hidden_states.reshape(B * S, H).contiguous()
        m = media.reshape(B * Sk, H).contiguous()
        ca = self.cross_attn
        h = self.input_layernorm(x)
        q, k, v = ca.q_proj(h), ca.k_proj(m), ca.v_proj(m)
        o = A.cross_attention(q, k, v, B=B, Sq=S, Sk=Sk, H=ca.heads, D=ca.head_dim, scale=ca.head_dim ** -0.5, krange=key_range)
        a = ca.o_proj(o)
        x = A.gated_residual(x, a, self.alpha_cross_attn.reshape(-1), gate.reshape(-1).contiguous())
        h = self.post_attention_layernorm(x)
        # gate|up as one GEMM: the two weights are concatenated on the fly (plumbing; a fused parameter would avoid the copy)
        gu = A.linear(h, torch.cat([self.mlp.gate_proj.weight, self.mlp.up_proj.weight], dim=0))
        d = self.mlp.down_proj(A.silu_mul(gu))
        x = A.gated_residual(x, d, self.alpha_dense.reshape(-1), None)
        return x.view(B, S, H)


def media_key_ranges(media_marker_positions, S, n_latents):
    """text position -> [begin, end) of the latents of the most recent clip marker at or before it; gate = 0 before any clip.
    media_marker_positions: list (per sample) of sorted marker positions.  Host-side index plumbing."""
    B = len(media_marker_positions)
    kr = torch.zeros((B, S, 2), dtype=torch.int32)
    gate = torch.zeros((B, S), dtype=torch.int32)
    for b, marks in enumerate(media_marker_positions):
        for ci, pos in enumerate(marks):
            end = marks[ci + 1] if ci + 1 < len(marks) else S
            kr[b, pos:end, 0] = ci * n_latents
            kr[b, pos:end, 1] = (ci + 1) * n_latents
            gate[b, pos:end] = 1
    return kr, gate
