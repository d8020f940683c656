"""BASELINE config 4 as a WORKLOAD: an AF1/AF2-style in-context-learning training step on the MI355X kernels.

    4 few-shot clips per sample -> synthetic audio-encoder features [B, 4, T_enc, D_enc]
      -> Linear(D_enc -> H) -> Perceiver resampler (64 latents per clip)                                   (flamingo.PerceiverResampler)
      -> media [B, 4*64, H]
    text ids with 4 <audio> markers -> embed -> decoder stack (Qwen2-style layers: RMSNorm, fused q|k|v + RoPE, causal GQA attention, SwiGLU)
      with a tanh-gated cross-attention block in front of every `xattn_every`-th layer: each text token attends to the latents of the
      most recent clip marker at or before it                                                              (flamingo.GatedCrossAttentionBlock)
      -> final RMSNorm -> lm_head + shifted cross-entropy on the labelled rows.

PARITY STATUS: **UNPINNED** with respect to AF1/AF2 (their OpenFlamingo-fork source is not mounted, no port exists in the wheels: SURVEY.md
§0, §8c).  The conditioning blocks follow HF's Flamingo re-implementation (Idefics), the decoder layers follow Qwen2 (AF2's LLM family); the
shapes below are BUILDER-DECLARED (SURVEY.md §8d "config 4").  The assembled step is checked against the CPU restatement
oracle/flamingo_oracle.py::icl_forward (tests/test_flamingo_gpu.py) - that pins the kernels and the wiring, not AF1/AF2's numerics.
Arithmetic: afk kernels through autograd_ops (gradients returned as tensors); the optimizer is the same fused AdamW kernel, per tensor.
"""
from __future__ import annotations

import torch
from torch import nn

from . import autograd_ops as A
from . import ops
from .flamingo import BF, GatedCrossAttentionBlock, PerceiverResampler, _Lin, _ones, _p, _zeros

# builder-declared shapes of the benchmark ("icl4"): AF2-like - a Qwen2.5-3B-class decoder, CLAP-sized encoder features
ICL4 = dict(vocab=151936, hidden=2048, inter=11008, layers=36, heads=16, kv_heads=2, head_dim=128, rms_eps=1e-6, rope_theta=1e6,
            xattn_every=4, xattn_heads=16, xattn_inter=8192, n_latents=64, resampler_depth=6, resampler_heads=16, resampler_head_dim=128,
            enc_dim=768, enc_frames=64, clips=4, audio_marker_id=151665)


class DecoderLayer(nn.Module):
    """Qwen2DecoderLayer (transformers/models/qwen2/modeling_qwen2.py:258-298) on the general-purpose autograd ops"""

    def __init__(self, c, device):
        super().__init__()
        H, I, nq, nkv = c["hidden"], c["inter"], c["heads"] * c["head_dim"], c["kv_heads"] * c["head_dim"]
        self.c = c
        self.input_layernorm = nn.Module()
        self.input_layernorm.weight = _ones(H, device)
        self.post_attention_layernorm = nn.Module()
        self.post_attention_layernorm.weight = _ones(H, device)
        self.qkv = nn.Module()
        self.qkv.weight, self.qkv.bias = _p(nq + 2 * nkv, H, device=device), _zeros(nq + 2 * nkv, device)
        self.o_proj, self.gate_up, self.down_proj = _Lin(nq, H, device), _Lin(H, 2 * I, device), _Lin(I, H, device)

    def forward(self, x, B, S, cos, sin):
        c = self.c
        h = A.rms_norm(x, self.input_layernorm.weight, c["rms_eps"])
        qkv = A.linear(h, self.qkv.weight, self.qkv.bias)
        qkv = A.rope(qkv, cos, sin, S=S, nheads=c["heads"] + c["kv_heads"], D=c["head_dim"])
        o = A.fused_self_attention(qkv, B=B, S=S, Hq=c["heads"], Hkv=c["kv_heads"], D=c["head_dim"], scale=c["head_dim"] ** -0.5)
        x = self.o_proj(o, residual=x)
        h = A.rms_norm(x, self.post_attention_layernorm.weight, c["rms_eps"])
        return self.down_proj(A.silu_mul(self.gate_up(h)), residual=x)


class FlamingoICLForCausalLM(nn.Module):
    def __init__(self, cfg: dict, device="cuda", seed: int = 0):
        super().__init__()
        torch.manual_seed(seed)
        c = self.c = dict(cfg)
        H = c["hidden"]
        self.device_ = torch.device(device)
        self.audio_proj = _Lin(c["enc_dim"], H, device)
        self.resampler = PerceiverResampler(H, c["resampler_depth"], c["resampler_heads"], c["resampler_head_dim"], c["n_latents"], device=device)
        self.embed_tokens = nn.Module()
        self.embed_tokens.weight = _p(c["vocab"], H, device=device)
        self.layers = nn.ModuleList([DecoderLayer(c, device) for _ in range(c["layers"])])
        self.xattn = nn.ModuleDict({str(i): GatedCrossAttentionBlock(H, c["xattn_heads"], c["xattn_inter"], eps=c["rms_eps"], device=device)
                                    for i in range(0, c["layers"], c["xattn_every"])})
        self.norm = nn.Module()
        self.norm.weight = _ones(H, device)
        self.lm_head = _Lin(H, c["vocab"], device)
        self._rope = {}

    def _rope_tables(self, S):
        if S not in self._rope:
            D = self.c["head_dim"]
            inv = 1.0 / (self.c["rope_theta"] ** (torch.arange(0, D, 2, device=self.device_, dtype=torch.float32) / D))
            fr = torch.arange(S, device=self.device_, dtype=torch.float32)[:, None] * inv[None, :]
            emb = torch.cat([fr, fr], dim=-1)
            self._rope[S] = (emb.cos().to(BF).contiguous(), emb.sin().to(BF).contiguous())
        return self._rope[S]

    def media_ranges(self, ids):
        """per text position: key interval = latents of the most recent <audio> marker at or before it; gate 0 before any marker.
        Integer index plumbing on the device (no sync)."""
        L = self.c["n_latents"]
        ci = (ids == self.c["audio_marker_id"]).to(torch.int32).cumsum(-1) - 1          # clip index, -1 before the first marker
        gate = (ci >= 0).to(torch.int32)
        lo = ci.clamp_min(0) * L
        kr = torch.stack([lo, torch.where(ci >= 0, lo + L, lo)], -1).to(torch.int32).contiguous()
        return kr, gate.contiguous()

    def forward(self, input_ids, audio_features, labels=None, label_rows=None):
        """input_ids int64 [B, S]; audio_features bf16 [B, clips, T_enc, D_enc]; labels int64 [B, S] (-100 = ignore);
        label_rows: optional precomputed int64 indices of the rows whose SHIFTED label is not -100 (avoids a host sync)"""
        c = self.c
        B, S = input_ids.shape
        nc, T, De = audio_features.shape[1:]
        H = c["hidden"]
        f = self.audio_proj(audio_features.reshape(B * nc * T, De).to(BF).contiguous())
        media = self.resampler(f.view(B * nc, T, H)).reshape(B, nc * c["n_latents"], H)
        kr, gate = self.media_ranges(input_ids)
        cos, sin = self._rope_tables(S)
        x = A.embedding(input_ids.reshape(-1).contiguous(), self.embed_tokens.weight)
        for i, layer in enumerate(self.layers):
            if str(i) in self.xattn:
                x = self.xattn[str(i)](x.view(B, S, H), media, kr, gate).reshape(B * S, H)
            x = layer(x, B, S, cos, sin)
        x = A.rms_norm(x, self.norm.weight, c["rms_eps"])
        if labels is None:
            return A.linear(x, self.lm_head.weight).view(B, S, -1)
        shift = torch.nn.functional.pad(labels, (0, 1), value=-100)[:, 1:].reshape(-1).contiguous()
        return A.lm_head_loss(x, self.lm_head.weight, shift, label_rows)


class TensorAdamW:
    """the fused AdamW kernel (afk_adamw_step: bf16 parameter + fp32 master / m / v) applied tensor by tensor - for models whose parameters
    are ordinary nn.Parameters rather than views of one arena"""

    def __init__(self, params, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.state = [(p.data.float().reshape(-1).clone(), torch.zeros(p.numel(), device=p.device), torch.zeros(p.numel(), device=p.device))
                      for p in self.params]
        self.t = 0

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def step(self):
        self.t += 1
        for p, (master, m, v) in zip(self.params, self.state):
            if p.grad is None:
                continue
            ops.adamw_step(master, m, v, p.grad.contiguous().reshape(-1), p.data.reshape(-1), lr=self.lr, beta1=self.betas[0], beta2=self.betas[1],
                           eps=self.eps, weight_decay=self.wd if p.dim() > 1 else 0.0, step=self.t)
