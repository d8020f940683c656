"""MI355X-native Audio Flamingo 3 model with the surface of transformers' AudioFlamingo3ForConditionalGeneration.

Mirrors (same constructor config classes, same ``forward`` argument names, same ``state_dict`` keys, ``generate``):
    transformers/models/audioflamingo3/modeling_audioflamingo3.py:570-642 (ForConditionalGeneration)
    :448-562 (Model), :302-416 (Encoder), :419-439 (projector); transformers/models/qwen2/modeling_qwen2.py:321-402.
The arithmetic is entirely the hand-written gfx950 kernels of libafk.so (functional.py); this module only owns
the parameter layout (arena.py) and the order of the stages.  Parameters named as in the oracle are views into
fused arena blocks (q|k|v and gate|up are stored contiguously so that each is one GEMM), so
``load_state_dict`` from an oracle checkpoint and ``state_dict`` round-trip unchanged.

Deviations from the oracle surface (documented, not silent):
  * training forward with ``labels`` fuses lm_head + loss; ``output.logits`` is still there, as in the reference, but LAZY: the [B, S, V]
    tensor is built on first access from the DETACHED final hidden states under ``no_grad`` - it carries NO gradient (the reference's is in
    the graph: an auxiliary loss built on ``outputs.logits`` must use ``forward(return_logits=True)``, which builds the logits eagerly inside
    the step's autograd graph) and reading it after an optimizer step raises (the weights are no longer the ones of ``output.loss``); lm_head
    and the loss run only on the rows whose shifted label is not -100 (identical loss and gradients);
  * ``gradient_checkpointing_enable()`` = every layer recomputed, as in the reference; ``gradient_checkpointing_kwargs={"policy": "budget"}``
    opts into recomputing only what a memory budget requires;
  * greedy token selection on the device (``afk_decode_select_greedy``) ignores NaN logits (picks the largest finite one) where ``torch.argmax``
    returns the NaN's index: a model that produces NaNs decodes differently - by design, it keeps the graph-replayed step free of host checks;
  * ``AFK_EXACT_FP32=1`` routes the inference forward / greedy ``generate`` through the exact fp32 verification kernels (exact.py);
  * ``forward(use_cache=True)`` / ``forward(past_key_values=cache)``: the reference's cache protocol for inference (prefill returns an
    ``AfkKVCache``, later calls append one or several tokens); same kernels and cache layout as ``generate``;
  * ``generate``: prefill fills a KV cache, each new token is one HIP-graph replay; greedy by default, ``do_sample=True`` with
    ``temperature`` / ``top_k`` / ``top_p`` / ``seed`` (the reference's logits-warper order), or ``num_beams > 1`` (beam search with the
    reference's scoring); ``generation_config`` supplies defaults; constrained / assisted decoding and custom logits processors are not built;
  * ``attention_mask`` rows must be one contiguous run of ones (left padding - the reference processor's default -, right padding, or
    both); masks with holes raise.  Hidden states of padded positions are zeros-attended garbage in both implementations and are
    never compared.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
from torch import nn

from . import _lib, ops
from ._lib import AfkError
from .arena import Arena
from . import functional as F_


class AfkKVCache:
    """KV cache handed between forward(use_cache=True) calls (the role of transformers' DynamicCache): per-layer post-RoPE keys
    K [L, B, Smax, Hkv*D], values stored transposed Vt [L, B, Hkv, D, pad64(Smax)] (the layout the decode attention kernels read), the
    number of filled positions and, for left-padded prompts, the first real position of every sample."""

    def __init__(self, K, Vt, lo, length):
        self.K, self.Vt, self.lo, self.length = K, Vt, lo, int(length)

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return self.length

    # ---- interop with the reference's cache object (TF/cache_utils.py DynamicCache; generation/utils.py:519-640 hands one to every forward)
    @staticmethod
    def is_reference_cache(obj) -> bool:
        return obj is not None and not isinstance(obj, AfkKVCache) and hasattr(obj, "layers") and hasattr(obj, "update") and hasattr(obj, "get_seq_length")

    @classmethod
    def adopt(cls, ref_cache, n_layers: int, Hkv: int, D: int, n_new: int, headroom: int, device, attention_mask=None):
        """the tensors of a reference DynamicCache (per layer: keys / values [B, Hkv, S, D], keys post-RoPE - Qwen2Attention.forward
        modeling_qwen2.py:213-214) re-laid into this cache: K [L, B, Smax, Hkv * D], values transposed Vt [L, B, Hkv, D, pad64(Smax)].  Left padding is
        read off the attention_mask of the call (which covers past + new positions, as the reference requires)."""
        from . import ops

        start = int(ref_cache.get_seq_length())
        if start == 0:
            return None   # an EMPTY reference cache on the prefill call (`past_key_values=DynamicCache()`): nothing to adopt
        if len(ref_cache.layers) != n_layers:
            raise AfkError(f"forward(past_key_values=<reference cache>): it holds {len(ref_cache.layers)} layers, the decoder has {n_layers}")
        k0 = ref_cache.layers[0].keys
        B = k0.shape[0]
        if tuple(k0.shape[1:]) != (Hkv, start, D):
            raise AfkError(f"forward(past_key_values=<reference cache>): layer 0 keys are {tuple(k0.shape)}, expected [B, {Hkv}, {start}, {D}]")
        Smax = start + n_new + headroom
        K = torch.zeros((n_layers, B, Smax, Hkv * D), device=device, dtype=torch.bfloat16)
        Vt = torch.zeros((n_layers, B, Hkv, D, ops.pad64(Smax)), device=device, dtype=torch.bfloat16)
        for i, layer in enumerate(ref_cache.layers):
            K[i, :, :start].copy_(layer.keys.to(device, torch.bfloat16).permute(0, 2, 1, 3).reshape(B, start, Hkv * D))
            Vt[i, :, :, :, :start].copy_(layer.values.to(device, torch.bfloat16).permute(0, 1, 3, 2))
        lo = torch.zeros(B, device=device, dtype=torch.int32)
        if attention_mask is not None:
            am = attention_mask.to(device)
            if am.shape[1] >= start and not bool(am[:, :start].all()):
                lo = (start - am[:, :start].sum(-1)).to(torch.int32)   # left padding: leading zeros of the past part
        return cls(K, Vt, lo, start)

    def write_back(self, ref_cache, first: int, n: int, Hkv: int, D: int):
        """append positions [first, first + n) of every layer to the reference cache the caller handed in (DynamicCache.update), so that it stays the
        caller's single source of truth: the next call - to this model or to the reference - finds past + new in it"""
        L, B = self.K.shape[0], self.K.shape[1]
        for i in range(L):
            k = self.K[i, :, first: first + n].reshape(B, n, Hkv, D).permute(0, 2, 1, 3).contiguous()
            v = self.Vt[i, :, :, :, first: first + n].permute(0, 1, 3, 2).contiguous()
            ref_cache.update(k, v, i)
        return ref_cache


class AF3Output:
    """The reference's ModelOutput surface (AudioFlamingo3CausalLMOutputWithPast, modeling_audioflamingo3.py:635-642): attribute, key, index and
    slice access, ``to_tuple()``, ``keys()`` / ``items()``.

    ``logits`` is ALWAYS available, as in the reference (which returns it whenever it returns a loss, :625-642) - but when ``labels`` were
    given the fused lm_head + loss path never built the [B, S, V] tensor, so it is materialised on FIRST ACCESS (one lm_head GEMM on the
    final hidden states the output keeps): a training loop that only reads ``out.loss`` / ``out[0]`` never pays for it, ``Trainer.evaluate``
    / ``compute_metrics`` (which read ``outputs[1:]``) get what they expect."""

    _FIELDS = ("loss", "logits", "past_key_values", "hidden_states", "attentions", "audio_hidden_states")

    def __init__(self, loss=None, logits=None, past_key_values=None, hidden_states=None, attentions=None, audio_hidden_states=None, logits_fn=None):
        self.loss, self._logits, self._logits_fn = loss, logits, logits_fn
        self.past_key_values, self.hidden_states, self.attentions, self.audio_hidden_states = past_key_values, hidden_states, attentions, audio_hidden_states

    @property
    def logits(self):
        if self._logits is None and self._logits_fn is not None:
            self._logits, self._logits_fn = self._logits_fn(), None
        return self._logits

    @logits.setter
    def logits(self, v):
        self._logits, self._logits_fn = v, None

    @property
    def logits_materialized(self) -> bool:
        return self._logits is not None

    def _present(self, name):
        return (self._logits is not None or self._logits_fn is not None) if name == "logits" else getattr(self, name) is not None

    def keys(self):
        return [f for f in self._FIELDS if self._present(f)]

    def items(self):
        return [(f, getattr(self, f)) for f in self.keys()]

    def to_tuple(self):
        return tuple(getattr(self, f) for f in self.keys())

    def get(self, k, default=None):
        return getattr(self, k) if k in self._FIELDS and self._present(k) else default

    def __contains__(self, k):
        return k in self._FIELDS and self._present(k)

    def __getitem__(self, k):
        if isinstance(k, str):
            if k not in self._FIELDS:
                raise KeyError(k)
            return getattr(self, k)
        if isinstance(k, int):
            return getattr(self, self.keys()[k])   # out[0] is the loss: nothing else is touched (no logits materialisation)
        return self.to_tuple()[k]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())


class _Holder(nn.Module):
    """parameter container that reproduces the oracle's module tree (and therefore its state_dict keys)"""


def _attach(root: nn.Module, name: str, param: nn.Parameter):
    parts = name.split(".")
    m = root
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, _Holder())
        m = m._modules[p]
    m.register_parameter(parts[-1], param)


class AudioFlamingo3ForConditionalGeneration(nn.Module):
    """Drop-in for the oracle class of the same name (config = transformers.AudioFlamingo3Config or a duck-typed object)."""

    def __init__(self, config, device="cuda", init_seed: Optional[int] = 0):
        super().__init__()
        self.config = config
        ac, tc = config.audio_config, config.text_config
        self.device_ = torch.device(device)
        # NOTE: construction on "cpu" is allowed for host-side logic (layout, state_dict, DP bucketing tests);
        # every compute entry point raises AfkError for non-HIP tensors - there is no CPU arithmetic path.
        self.E, self.enc_heads, self.enc_ffn = ac.hidden_size, ac.num_attention_heads, ac.intermediate_size
        self.enc_layers, self.n_mels, self.max_pos = ac.num_hidden_layers, ac.num_mel_bins, ac.max_source_positions
        self.H, self.I, self.V = tc.hidden_size, tc.intermediate_size, tc.vocab_size
        self.Hq, self.Hkv = tc.num_attention_heads, tc.num_key_value_heads
        self.D = getattr(tc, "head_dim", None) or self.H // self.Hq
        self.dec_layers, self.rms_eps = tc.num_hidden_layers, tc.rms_norm_eps
        rp = getattr(tc, "rope_parameters", None) or {}
        self.rope_theta = float(rp.get("rope_theta", getattr(tc, "rope_theta", 10000.0)) if isinstance(rp, dict) else getattr(tc, "rope_theta", 10000.0))
        self.audio_token_id = config.audio_token_id
        self.check_placeholders = True
        for name, v in (("audio hidden", self.E), ("encoder ffn", self.enc_ffn), ("text hidden", self.H), ("intermediate", self.I),
                        ("vocab", self.V), ("3*n_mels", 3 * self.n_mels)):
            if v % 64:
                raise AfkError(f"{name} size {v} must be a multiple of 64 for the MFMA GEMM")
        if (self.E // self.enc_heads) not in (32, 64, 128) or self.D not in (32, 64, 128):
            raise AfkError("head_dim must be 32, 64 or 128")

        # ---------------- arena layout (forward order; one bucket per layer)
        a = Arena(self.device_)
        self.arena = a
        E, Fd, H, I, V = self.E, self.enc_ffn, self.H, self.I, self.V
        at, pj, lm = "model.audio_tower.", "model.multi_modal_projector.", "model.language_model."
        b = a.new_bucket("stem")
        a.add(at + "conv1.weight", (E, self.n_mels, 3), b, shadow="conv")
        a.add(at + "conv1.bias", (E,), b, decay=False)
        a.add(at + "conv2.weight", (E, E, 3), b, shadow="conv")
        a.add(at + "conv2.bias", (E,), b, decay=False)
        for i in range(self.enc_layers):
            p = f"{at}layers.{i}."
            b = a.new_bucket(f"enc{i}")
            a.add(p + "self_attn_layer_norm.weight", (E,), b, decay=False)
            a.add(p + "self_attn_layer_norm.bias", (E,), b, decay=False)
            a.add(p + "self_attn.qkv.weight", (3 * E, E), b, shadow="T")
            a.add(p + "self_attn.qkv.bias", (3 * E,), b, decay=False)
            a.add(p + "self_attn.out_proj.weight", (E, E), b, shadow="T")
            a.add(p + "self_attn.out_proj.bias", (E,), b, decay=False)
            a.add(p + "final_layer_norm.weight", (E,), b, decay=False)
            a.add(p + "final_layer_norm.bias", (E,), b, decay=False)
            a.add(p + "fc1.weight", (Fd, E), b, shadow="T")
            a.add(p + "fc1.bias", (Fd,), b, decay=False)
            a.add(p + "fc2.weight", (E, Fd), b, shadow="T")
            a.add(p + "fc2.bias", (E,), b, decay=False)
        b = a.new_bucket("enc_out")
        a.add(at + "layer_norm.weight", (E,), b, decay=False)
        a.add(at + "layer_norm.bias", (E,), b, decay=False)
        a.add(pj + "linear_1.weight", (H, E), b, shadow="T")
        a.add(pj + "linear_1.bias", (H,), b, decay=False)
        a.add(pj + "linear_2.weight", (H, H), b, shadow="T")
        a.add(pj + "linear_2.bias", (H,), b, decay=False)
        b = a.new_bucket("embed")
        a.add(lm + "embed_tokens.weight", (V, H), b)
        nq, nkv = self.Hq * self.D, self.Hkv * self.D
        for i in range(self.dec_layers):
            p = f"{lm}layers.{i}."
            b = a.new_bucket(f"dec{i}")
            a.add(p + "input_layernorm.weight", (H,), b, decay=False)
            a.add(p + "self_attn.qkv.weight", (nq + 2 * nkv, H), b, shadow="T")
            a.add(p + "self_attn.qkv.bias", (nq + 2 * nkv,), b, decay=False)
            a.add(p + "self_attn.o_proj.weight", (H, nq), b, shadow="T")
            a.add(p + "post_attention_layernorm.weight", (H,), b, decay=False)
            a.add(p + "mlp.gate_up.weight", (2 * I, H), b, shadow="T")
            a.add(p + "mlp.down_proj.weight", (H, I), b, shadow="T")
        b = a.new_bucket("head")
        a.add(lm + "norm.weight", (H,), b, decay=False)
        a.add("lm_head.weight", (V, H), b, shadow="T")
        a.finalize()
        # weights whose dgrad reads W itself (functional.NN_DGRAD_SUFFIXES: the NN kernel) never need an eager W^T shadow: not allocated, not refreshed
        for blk in a.order:
            if blk.shadow_kind == "T" and F_.NN_DGRAD_SUFFIXES and blk.key.endswith(F_.NN_DGRAD_SUFFIXES):
                blk.shadow_lazy = True

        # ---------------- oracle-named parameters = views into the arena
        self._prm = {}
        def P(name, blk_key, rows=None):
            blk = a[blk_key]
            d, g = (blk.data, blk.grad) if rows is None else (blk.data[rows[0]: rows[1]], blk.grad[rows[0]: rows[1]])
            prm = nn.Parameter(d)
            prm.grad = g
            self._prm[name] = prm
            _attach(self, name, prm)

        for k in ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "layer_norm.weight", "layer_norm.bias"):
            P(at + k, at + k)
        # frozen sinusoid table (requires_grad False in the oracle, :332) lives outside the arena
        pos_prm = nn.Parameter(torch.zeros((self.max_pos, E), device=self.device_, dtype=torch.bfloat16), requires_grad=False)
        _attach(self, at + "embed_positions.weight", pos_prm)
        self._prm[at + "embed_positions.weight"] = pos_prm
        a.extra_state.append(pos_prm.data)
        for i in range(self.enc_layers):
            p = f"{at}layers.{i}."
            P(p + "self_attn.q_proj.weight", p + "self_attn.qkv.weight", (0, E))
            P(p + "self_attn.k_proj.weight", p + "self_attn.qkv.weight", (E, 2 * E))
            P(p + "self_attn.v_proj.weight", p + "self_attn.qkv.weight", (2 * E, 3 * E))
            P(p + "self_attn.q_proj.bias", p + "self_attn.qkv.bias", (0, E))
            P(p + "self_attn.v_proj.bias", p + "self_attn.qkv.bias", (2 * E, 3 * E))
            for k in ("self_attn.out_proj.weight", "self_attn.out_proj.bias", "self_attn_layer_norm.weight", "self_attn_layer_norm.bias",
                      "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "final_layer_norm.weight", "final_layer_norm.bias"):
                P(p + k, p + k)
        for k in ("linear_1.weight", "linear_1.bias", "linear_2.weight", "linear_2.bias"):
            P(pj + k, pj + k)
        P(lm + "embed_tokens.weight", lm + "embed_tokens.weight")
        for i in range(self.dec_layers):
            p = f"{lm}layers.{i}."
            for nm, (s, e) in (("q_proj", (0, nq)), ("k_proj", (nq, nq + nkv)), ("v_proj", (nq + nkv, nq + 2 * nkv))):
                P(p + f"self_attn.{nm}.weight", p + "self_attn.qkv.weight", (s, e))
                P(p + f"self_attn.{nm}.bias", p + "self_attn.qkv.bias", (s, e))
            P(p + "self_attn.o_proj.weight", p + "self_attn.o_proj.weight")
            P(p + "mlp.gate_proj.weight", p + "mlp.gate_up.weight", (0, I))
            P(p + "mlp.up_proj.weight", p + "mlp.gate_up.weight", (I, 2 * I))
            P(p + "mlp.down_proj.weight", p + "mlp.down_proj.weight")
            P(p + "input_layernorm.weight", p + "input_layernorm.weight")
            P(p + "post_attention_layernorm.weight", p + "post_attention_layernorm.weight")
        P(lm + "norm.weight", lm + "norm.weight")
        P("lm_head.weight", "lm_head.weight")

        self._at, self._pj, self._lm = at, pj, lm
        self._rope_cache = {}
        self.gradient_checkpointing = False
        if init_seed is not None:
            self.init_weights(init_seed)
        self.training = True

    # ------------------------------------------------------------------ init (oracle _init_weights: N(0, initializer_range), norms 1/0, biases 0)
    @torch.no_grad()
    def init_weights(self, seed: int = 0):
        a = self.arena
        std = float(getattr(self.config.audio_config, "initializer_range", 0.02))
        a.init_normal_(std, seed)
        for blk in a.order:
            if blk.key.endswith("norm.weight"):
                blk.data.fill_(1.0)
            elif blk.key.endswith(".bias"):
                blk.data.zero_()
        g = torch.Generator(device=self.device_)
        g.manual_seed(seed + 1)
        self.embed_positions.data.copy_((torch.randn(self.embed_positions.shape, device=self.device_, generator=g) * std).to(torch.bfloat16))
        a.step_counter += 1

    def load_state_dict(self, sd, strict=True, assign=False):
        r = super().load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()}, strict=strict)
        # the k-third of the fused encoder bias is not a parameter (k_proj has no bias): keep it zero
        E = self.E
        for i in range(self.enc_layers):
            self.arena[f"{self._at}layers.{i}.self_attn.qkv.bias"].data[E: 2 * E].zero_()
        self.arena.step_counter += 1  # invalidate W^T shadows
        return r

    def trainable_numel(self):
        return sum(b.numel for b in self.arena.order)

    # ------------------------------------------------------------------ small PreTrainedModel conveniences scripts rely on
    @property
    def device(self):
        return self.device_

    @property
    def dtype(self):
        return torch.bfloat16

    def num_parameters(self, only_trainable: bool = False, exclude_embeddings: bool = False) -> int:
        """PreTrainedModel.num_parameters (the HF Trainer calls it with exclude_embeddings=True for its FLOPs estimate)"""
        skip = {id(self._prm[self._lm + "embed_tokens.weight"]), id(self.embed_positions)} if exclude_embeddings else set()
        return sum(p.numel() for p in self.parameters() if (p.requires_grad or not only_trainable) and id(p) not in skip)

    def get_input_embeddings(self):
        """holder module whose `.weight` is the arena view of embed_tokens (the reference returns its nn.Embedding)"""
        return self.model.language_model.embed_tokens

    def get_output_embeddings(self):
        return self.lm_head

    # ------------------------------------------------------------------ checkpoints (PreTrainedModel.from_pretrained / save_pretrained surface)
    @classmethod
    def from_pretrained(cls, path, device="cuda", **kwargs):
        """Load a reference checkpoint directory (what AudioFlamingo3ForConditionalGeneration.save_pretrained writes / what
        `nvidia/audio-flamingo-3-hf` unpacks to): config.json + *.safetensors (single file or sharded with model.safetensors.index.json).
        The parameter names are the reference's, so the tensors drop straight into the arena views."""
        import json
        import os

        from safetensors import safe_open
        from transformers import AutoConfig

        config = AutoConfig.from_pretrained(path)
        model = cls(config, device=device, init_seed=None, **kwargs)
        idx = os.path.join(path, "model.safetensors.index.json")
        files = sorted(set(json.load(open(idx))["weight_map"].values())) if os.path.exists(idx) else ["model.safetensors"]
        sd = {}
        for fn in files:
            with safe_open(os.path.join(path, fn), framework="pt", device="cpu") as f:
                for k in f.keys():
                    sd[cls._runtime_key(k)] = f.get_tensor(k)
        model.load_state_dict(sd, strict=True)
        return model

    # checkpoint files use the hub layout; the reference renames on load (transformers/conversion_mapping.py "qwen2_audio":
    # ^language_model.model -> model.language_model, ^language_model.lm_head -> lm_head, ^audio_tower -> model.audio_tower,
    # ^multi_modal_projector -> model.multi_modal_projector) and reverses it on save
    _HUB_PREFIXES = (("language_model.model.model.", "model.language_model."), ("language_model.model.", "model.language_model."),
                     ("language_model.lm_head.", "lm_head."), ("audio_tower.", "model.audio_tower."),
                     ("multi_modal_projector.", "model.multi_modal_projector."))

    @classmethod
    def _runtime_key(cls, k: str) -> str:
        for src, dst in cls._HUB_PREFIXES:
            if k.startswith(src):
                return dst + k[len(src):]
        return k

    @classmethod
    def _hub_key(cls, k: str) -> str:
        for src, dst in cls._HUB_PREFIXES[1:]:
            if k.startswith(dst):
                return src + k[len(dst):]
        return k

    def save_pretrained(self, path, max_shard_size: int = 5 << 30):
        """config.json + safetensors shards with the reference's parameter names (loadable by the reference's from_pretrained)"""
        import json
        import os

        from safetensors.torch import save_file

        os.makedirs(path, exist_ok=True)
        self.config.save_pretrained(path)
        shards, cur, size = [], {}, 0
        for k, v in self.state_dict().items():
            t = v.detach().to("cpu").contiguous().clone()
            nb = t.numel() * t.element_size()
            if cur and size + nb > max_shard_size:
                shards.append(cur)
                cur, size = {}, 0
            cur[self._hub_key(k)] = t
            size += nb
        shards.append(cur)
        if len(shards) == 1:
            save_file(shards[0], os.path.join(path, "model.safetensors"), metadata={"format": "pt"})
            return
        wm = {}
        for i, sh in enumerate(shards):
            fn = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
            save_file(sh, os.path.join(path, fn), metadata={"format": "pt"})
            wm.update({k: fn for k in sh})
        json.dump({"metadata": {"total_size": sum(t.numel() * t.element_size() for sh in shards for t in sh.values())}, "weight_map": wm},
                  open(os.path.join(path, "model.safetensors.index.json"), "w"))

    # ------------------------------------------------------------------ helpers
    def _rope_tables(self, S: int):
        """Qwen2RotaryEmbedding.forward (modeling_qwen2.py:91-102): fp32 angles, cos/sin rounded to bf16"""
        if S not in self._rope_cache:
            D = self.D
            inv = 1.0 / (self.rope_theta ** (torch.arange(0, D, 2, device=self.device_, dtype=torch.float32) / D))
            fr = torch.arange(S, device=self.device_, dtype=torch.float32)[:, None] * inv[None, :]
            emb = torch.cat([fr, fr], dim=-1)
            self._rope_cache[S] = (emb.cos().to(torch.bfloat16).contiguous(), emb.sin().to(torch.bfloat16).contiguous())
        return self._rope_cache[S]

    # Activation checkpointing (oracle: GradientCheckpointingLayer.__call__, transformers/modeling_layers.py:79-114, switched on by
    # gradient_checkpointing_enable, modeling_utils.py:3187: EVERY transformer layer re-runs its forward in backward - the answer for 80 GB
    # parts).  gradient_checkpointing_enable() with no arguments means exactly that here too (policy "full", round 5 / ADVICE r04: a caller that
    # switches checkpointing on to FIT must get the reference's memory behaviour).  Opt-in (gradient_checkpointing_kwargs / env):
    #     policy = "budget" (env AFK_CKPT_POLICY)  recompute only what a memory budget requires: 288 GB of HBM3E hold the activations of a 5-minute clip
    #                                              outright (185 GiB), and a recomputed layer costs a third of its forward + backward time again
    #     memory_budget_gib = None                 -> CKPT_BUDGET_FRACTION of the device's memory (0.85 x 288 GiB = 245 GiB on MI355X); env AFK_CKPT_BUDGET_GIB
    # The budget plan - how many of the FIRST layers of each tower are recomputed (their recompute runs last in backward, when the later layers'
    # activations are already gone) - is made per forward from the batch geometry, what this process has allocated, what the DEVICE still has free
    # (hipMemGetInfo: other processes count) plus this process's own cached blocks, and the W^T shadows that will still be allocated lazily; gradients
    # are bit-identical whatever the plan (tests/test_model_gpu.py::test_gradient_checkpointing_matches).  Every new plan is logged.
    CKPT_BUDGET_FRACTION = 0.85

    def gradient_checkpointing_enable(self, gradient_checkpointing_kwargs=None):
        kw = dict(gradient_checkpointing_kwargs or {})
        self.gradient_checkpointing = True
        self.ckpt_policy = kw.get("policy", os.environ.get("AFK_CKPT_POLICY", "full"))
        b = kw.get("memory_budget_gib", os.environ.get("AFK_CKPT_BUDGET_GIB"))
        self.ckpt_budget_bytes = None if b is None else int(float(b) * 2 ** 30)
        if self.ckpt_policy not in ("budget", "full"):
            raise ValueError(f"gradient_checkpointing policy {self.ckpt_policy!r}: 'budget' or 'full'")

    def gradient_checkpointing_disable(self):
        self.gradient_checkpointing = False

    ckpt_policy, ckpt_budget_bytes, ckpt_plan, _ckpt_logged = "full", None, None, None

    def activation_bytes_per_layer(self, windows: int, dec_rows: int):
        """bytes one encoder layer / one decoder layer keeps for backward (functional.EncoderLayerFn / DecoderLayerFn.save_for_backward):
        encoder, per row of W x 1500: x, h, o, x2, h2 (E each), qkv (3E), fc1 pre-activation and GELU output (F each) + fp32 statistics;
        decoder, per token: x, h, o, x2, h2 (H), qkv ((Hq + 2 Hkv) D), gate|up (2 I), SwiGLU output (I) + fp32 lse per head and statistics"""
        ac, tc = self.config.audio_config, self.config.text_config
        E, Fi = ac.hidden_size, ac.intermediate_size
        enc = windows * self.max_pos * (2 * (8 * E + 2 * Fi) + 4 * (4 + self.enc_heads))
        H, I = tc.hidden_size, tc.intermediate_size
        dec = dec_rows * (2 * (5 * H + (self.Hq + 2 * self.Hkv) * self.D + 3 * I) + 4 * (2 + self.Hq))
        return enc, dec

    def plan_checkpointing(self, windows: int, dec_rows: int, allocated_bytes: int, total_bytes: int, usable_bytes: Optional[int] = None,
                           pending_bytes: int = 0):
        """-> {"enc": n, "dec": n, ...}: how many of the first layers of each tower are recomputed.  Pure host arithmetic (tested on the CPU).
        usable_bytes: what this process can still obtain (device-free memory + its own cached-but-unallocated blocks; None = unknown, trust the budget);
        pending_bytes: allocations that will happen later in the step outside the activations (W^T shadows created lazily by the first backward)."""
        enc_b, dec_b = self.activation_bytes_per_layer(windows, dec_rows)
        if self.ckpt_policy == "full":
            return {"enc": self.enc_layers if windows else 0, "dec": self.dec_layers, "policy": "full"}
        budget = self.ckpt_budget_bytes if self.ckpt_budget_bytes is not None else int(self.CKPT_BUDGET_FRACTION * total_bytes)
        # what backward needs on top of the kept activations: one layer being recomputed + one layer's worth of gradient temporaries in each tower
        # in flight (three streams), the lm_head logits chunk and its fp32 split-K partials, allocator slack
        V = self.config.text_config.vocab_size
        headroom = 2 * dec_b + 2 * enc_b + 3 * F_.LMHeadLossFn.CHUNK * V * 2 + (6 << 30)
        avail = budget - allocated_bytes
        if usable_bytes is not None:
            avail = min(avail, int(usable_bytes))
        avail -= headroom + int(pending_bytes)
        need = self.enc_layers * enc_b + self.dec_layers * dec_b
        n_dec = n_enc = 0
        if need > avail:
            n_dec = min(self.dec_layers, -(-(need - avail) // max(dec_b, 1))) if dec_b else 0
            need -= n_dec * dec_b
            if need > avail and enc_b:
                n_enc = min(self.enc_layers, -(-(need - avail) // enc_b))
        return {"enc": int(n_enc), "dec": int(n_dec), "policy": "budget", "budget_gib": round(budget / 2 ** 30, 1),
                "allocated_gib": round(allocated_bytes / 2 ** 30, 1), "usable_gib": None if usable_bytes is None else round(usable_bytes / 2 ** 30, 1),
                "pending_gib": round(pending_bytes / 2 ** 30, 2), "enc_layer_gib": round(enc_b / 2 ** 30, 3), "dec_layer_gib": round(dec_b / 2 ** 30, 3)}

    def _pending_shadow_bytes(self) -> int:
        """W^T / conv shadows that do not exist yet and will be allocated by the first backward that needs them"""
        return sum(2 * b.numel * (2 if b.shadow_kind == "conv" else 1) for b in self.arena.order if b.shadow_kind is not None and b.shadow is None)

    def _make_ckpt_plan(self, windows, dec_rows):
        if not (self.gradient_checkpointing and torch.is_grad_enabled()):
            self.ckpt_plan = None
            return
        free, total = torch.cuda.mem_get_info(self.device_)
        alloc, reserved = torch.cuda.memory_allocated(self.device_), torch.cuda.memory_reserved(self.device_)
        self.ckpt_plan = self.plan_checkpointing(windows, dec_rows, alloc, total, usable_bytes=free + max(reserved - alloc, 0),
                                                 pending_bytes=self._pending_shadow_bytes())
        key = (windows, dec_rows, self.ckpt_plan["enc"], self.ckpt_plan["dec"], self.ckpt_plan["policy"])
        if key != self._ckpt_logged:
            self._ckpt_logged = key
            import logging

            logging.getLogger("audio_flamingo_amd").info("activation checkpointing plan for %d windows / %d decoder rows: %s", windows, dec_rows, self.ckpt_plan)

    def _layer(self, fn, *args, ckpt=True):
        if ckpt and self.gradient_checkpointing and torch.is_grad_enabled():
            import contextlib

            from torch.utils.checkpoint import checkpoint

            # the second context manager wraps the RE-RUN of the layer inside backward: functional.py skips what only produces the (discarded) output there
            return checkpoint(fn, *args, use_reentrant=False, context_fn=lambda: (contextlib.nullcontext(), F_.recomputing()))
        return fn(*args)

    def _require_hip(self):
        if self.device_.type != "cuda" or not torch.cuda.is_available():
            raise AfkError("audio_flamingo_amd: forward needs a HIP device (MI355X); this package has no CPU arithmetic path")

    @property
    def embed_positions(self):
        return self._prm[self._at + "embed_positions.weight"]

    def _anchor(self, key):
        """a real nn.Parameter of the stage: makes the Function's output require grad (its own grad slot stays None)"""
        return self._prm[key]

    # ------------------------------------------------------------------ audio tower + projector (a2-a9)
    def _post_encoder(self, x, W, T3, n_tok, input_ids):
        """hook between the audio tower and the projector (identity for AF3; Music Flamingo rotates by time here)"""
        return x

    def get_audio_features(self, input_features, input_features_mask=None, input_ids=None):
        """-> (audio_rows [W*T3, H] (all rows, padded windows included), tokens_per_window int64[W] or None)"""
        self._require_hip()
        a, at = self.arena, self._at
        feats = input_features.contiguous()
        if not feats.is_cuda:
            raise AfkError("input_features must be on the HIP device")
        if feats.dtype not in (torch.float32, torch.bfloat16):
            feats = feats.float()
        W, C, T = feats.shape
        T2 = (T - 1) // 2 + 1
        if T2 != self.max_pos:
            raise AfkError(f"input_features must have {2 * self.max_pos} frames (got {T}); embed_positions add requires it (:385)")
        kv_len, n_tok = None, None
        if input_features_mask is not None:
            L0 = input_features_mask.to(self.device_).sum(-1).to(torch.int64)
            L1 = (L0 - 1) // 2 + 1
            n_tok = (L1 - 2) // 2 + 1
            kv_len = L1.to(torch.int32).contiguous()
        x = F_.ConvStemFn.apply(feats, self._anchor(at + "conv1.weight"), a,
                                (at + "conv1.weight", at + "conv1.bias", at + "conv2.weight", at + "conv2.bias"),
                                self.embed_positions.data, W, T, C)
        if self.ckpt_plan is None or self.ckpt_plan.get("_windows") != W:   # stand-alone call (forward() plans both towers before it gets here)
            self._make_ckpt_plan(W, 0)
        n_ck = self.enc_layers if self.ckpt_plan is None else self.ckpt_plan["enc"]
        for i in range(self.enc_layers):
            p = f"{at}layers.{i}."
            x = self._layer(F_.EncoderLayerFn.apply, x, self._anchor(p + "fc1.weight"), a, p, W, T2, self.enc_heads, kv_len, ckpt=i < n_ck)
        T3 = T2 // 2
        x = F_.PoolNormFn.apply(x, self._anchor(at + "layer_norm.weight"), a, at + "layer_norm.weight", at + "layer_norm.bias", W * T3)
        x = self._post_encoder(x, W, T3, n_tok, input_ids)
        x = F_.ProjectorFn.apply(x, self._anchor(self._pj + "linear_1.weight"), a, self._pj)
        return x, n_tok

    # ------------------------------------------------------------------ forward (a10-a18)
    @staticmethod
    def _mask_intervals(attention_mask):
        """attention_mask [B, S] of 0/1 -> (lo [B], hi [B]) of the single run of ones per row, or None when nothing is padded.
        Computed on the mask's own device: a CPU mask (what a collator hands over) costs no device sync."""
        am = attention_mask != 0
        if bool(am.all()):
            return None
        S = am.shape[1]
        n = am.sum(-1)
        lo = am.to(torch.uint8).argmax(-1)
        lo = torch.where(n > 0, lo, torch.zeros_like(lo))
        hi = lo + n
        ar = torch.arange(S, device=am.device)[None]
        if not bool((am == ((ar >= lo[:, None]) & (ar < hi[:, None]))).all()):
            raise AfkError("attention_mask rows must each be ONE contiguous run of ones (left and/or right padding); got a mask with holes")
        return lo, hi

    _rows_cache = None  # (weakref(labels tensor), version, rows) - see _valid_rows
    # True: the POSITIONS of the labelled rows never change for a given labels tensor object, only the token values written into it do
    # (a static input buffer refilled in place every step: bench.py's rotating synthetic batches, a HIP-graph-replayed step).  The row
    # indices are then remembered per tensor object regardless of its version - no device scan, no host sync per step.
    label_rows_static = False
    label_rows_check = True   # with label_rows_static: device-side check (no sync) that the labelled-row COUNT still matches; a mismatch turns the loss into NaN
    _rows_poison = None

    def _valid_rows(self, labels):
        """-> (shift_labels [B*S] on the device, rows) with rows = int64 indices (device) of the positions whose SHIFTED label is not -100,
        or None when the lm_head should run on every row (all valid, or none).  The row count must be known on the host (it sizes the
        GEMMs): labels that arrive on the CPU (a collator's output) are scanned there - no device sync; device labels are scanned on
        the device (one sync) and the result is remembered for as long as the SAME tensor object (same version) comes back."""
        import weakref

        dev = self.device_
        sh = torch.nn.functional.pad(labels, (0, 1), value=-100)[:, 1:].reshape(-1)
        if not labels.is_cuda:
            rows = (sh != -100).nonzero().reshape(-1)
            shift = sh.to(dev).contiguous()
            rows = rows.to(dev) if 0 < rows.numel() < sh.numel() else None
            return shift, rows
        c = self._rows_cache
        if c is not None and c[0]() is labels and (c[1] == labels._version or self.label_rows_static):
            if self.label_rows_static and c[1] != labels._version and self.label_rows_check:
                # the promise is checked without a host sync: the number of labelled positions is compared ON THE DEVICE with the remembered
                # count and a mismatch poisons the loss (NaN), so a batch whose labelled positions moved cannot train on stale rows silently
                n_now = (sh != -100).sum()
                self._rows_poison = torch.where(n_now == c[3], 0.0, float("nan")).to(torch.float32)
            return sh.contiguous(), c[2]
        rows = (sh != -100).nonzero().reshape(-1)  # host sync: the count sizes the GEMMs
        rows = rows if 0 < rows.numel() < sh.numel() else None
        # the REAL labelled count is always remembered (rows is None both for "no row" and "every row"; label_rows_static may be switched on after this call,
        # and a remembered -1 would poison the next in-place label update - ADVICE r04)
        self._rows_cache = (weakref.ref(labels), labels._version, rows, int(rows.numel()) if rows is not None else int((sh != -100).sum()))
        return sh.contiguous(), rows

    def forward(self, input_ids=None, input_features=None, input_features_mask=None, attention_mask=None, position_ids=None,
                past_key_values=None, inputs_embeds=None, labels=None, use_cache=None, logits_to_keep=0, return_logits=None,
                num_items_in_batch=None, **kwargs):
        if (input_ids is None) == (inputs_embeds is None):
            raise ValueError("You must specify exactly one of input_ids or inputs_embeds")
        self._require_hip()
        from . import exact as _exact

        if _exact.ENABLED and labels is None and inputs_embeds is None and past_key_values is None and not torch.is_grad_enabled():
            # AFK_EXACT_FP32=1 (verification mode, exact.py): the inference forward in exact fp32 on the afk_x32_* kernels - fp32 logits, no KV cache.
            # What the mode does not implement is refused, never silently dropped
            if position_ids is not None or use_cache or logits_to_keep not in (0, None):
                raise AfkError("AFK_EXACT_FP32=1: the exact-fp32 verification forward takes no position_ids / use_cache / logits_to_keep "
                               "(positions are cumsum(attention_mask) - 1, every row's logits are returned, there is no KV cache); unset AFK_EXACT_FP32 for those")
            return AF3Output(logits=_exact.logits(self, input_ids, input_features, input_features_mask, attention_mask))
        if past_key_values is not None or use_cache:
            # inference with the reference's cache protocol (modeling_qwen2.py:213-214, 360-364): prefill returns a cache, later calls
            # append their tokens to it.  Same kernels and cache layout as generate(); no autograd graph.
            if labels is not None:
                raise AfkError("forward(labels=..., use_cache=True): the KV-cache path is inference only")
            return self._forward_cached(input_ids, inputs_embeds, input_features, input_features_mask, attention_mask, past_key_values,
                                        logits_to_keep, position_ids)
        a, lm = self.arena, self._lm
        if inputs_embeds is not None:
            # the reference merges audio only when input_ids are given (modeling_audioflamingo3.py:532-545): precomputed embeddings pass through
            B, S = inputs_embeds.shape[:2]
            ids = ids_flat = None
            input_features = None
        else:
            ids = input_ids.to(self.device_)
            B, S = ids.shape
            ids_flat = ids.reshape(-1).contiguous()
        audio, src = None, None
        # checkpoint plan of BOTH towers from the batch geometry, before the first layer allocates anything (memory-budgeted recompute)
        self._make_ckpt_plan(0 if input_features is None else int(input_features.shape[0]), B * S)
        if self.ckpt_plan is not None:
            self.ckpt_plan["_windows"] = 0 if input_features is None else int(input_features.shape[0])
        if input_features is not None:
            audio, n_tok = self.get_audio_features(input_features.to(self.device_), input_features_mask, input_ids=ids)
            src, cnt = ops.placeholder_scan(ids_flat, self.audio_token_id)
            T3 = audio.shape[0] // input_features.shape[0]
            if n_tok is not None:
                # rank r among placeholders -> row (window, t) of the padded audio buffer  (:483-486 boolean index)
                csum = torch.cumsum(n_tok, 0)
                r = src.clamp_min(0).to(torch.int64)
                win = torch.searchsorted(csum, r, right=True).clamp_max(n_tok.numel() - 1)
                row = win * T3 + (r - (csum - n_tok)[win])
                src = torch.where(src >= 0, row.to(torch.int32), src).contiguous()
                expected = csum[-1]
            else:
                expected = audio.shape[0]
            if self.check_placeholders:
                n = int(cnt.item())
                e = int(expected)
                if n != e:
                    raise ValueError(f"Audio features and audio tokens do not match, tokens: {n}, features: {e}")
        kv_len, krange, kv_lo = None, None, None
        if attention_mask is not None:
            iv = self._mask_intervals(attention_mask)
            if iv is not None:
                lo, hi = iv
                if bool((lo == 0).all()):   # right padding only: the LDS-staged causal kernels take the key count per sample
                    kv_len = hi.to(self.device_, torch.int32).contiguous()
                elif ops._use_lds(self.D) and self.left_pad_on_lds_kernels:
                    # left padding (processing_audioflamingo3.py:46) on the same kernels: keys [lo_b, hi_b) per sample, causal
                    kv_len = hi.to(self.device_, torch.int32).contiguous()
                    kv_lo = lo.to(self.device_, torch.int32).contiguous()
                else:                        # other head sizes: causal AND key in [lo_b, hi_b) per query row on the interval kernels
                    lo, hi = lo.to(self.device_, torch.int32), hi.to(self.device_, torch.int32)
                    i1 = torch.arange(1, S + 1, device=self.device_, dtype=torch.int32)[None]
                    end = torch.minimum(i1, hi[:, None])
                    krange = torch.stack([lo[:, None].expand(B, S), torch.maximum(end, lo[:, None])], -1).contiguous()
        pos = None
        if position_ids is not None:
            pos = position_ids.to(self.device_).expand(B, S).reshape(-1).to(torch.int32).contiguous()
        cos, sin = self._rope_tables(S if pos is None else int(self.config.text_config.max_position_embeddings))
        if inputs_embeds is not None:
            x = inputs_embeds.to(self.device_, torch.bfloat16).reshape(B * S, self.H).contiguous()
        else:
            x = F_.EmbedScatterFn.apply(audio, self._anchor(lm + "embed_tokens.weight"), a, lm + "embed_tokens.weight", ids_flat, src)
        audio_hidden = audio
        if kwargs.get("output_attentions"):
            raise AfkError("output_attentions=True: the attention kernels never materialise the probabilities (the reference's default sdpa path returns none either)")
        want_hidden = bool(kwargs.get("output_hidden_states", getattr(self.config, "output_hidden_states", False)))
        hidden = [x.reshape(B, S, -1)] if want_hidden else None      # the reference's tuple: merged embeddings, every layer's output, ...
        # the positions the loss reads, known before the decoder runs: behind its attention, the LAST decoder layer (and the final norm) run on those rows alone
        # (round 6, DecoderLayerFn rows=...: a quarter of the rows on the 256-answer-token batches of the benchmark) whenever nobody asks for the hidden states or
        # the logits of the other positions inside this call; the lazy `output.logits` of a labelled call re-runs the last layer on every row when it is read
        shift = rows = None
        if labels is not None:
            shift, rows = self._valid_rows(labels) if self.loss_on_valid_rows_only else (
                torch.nn.functional.pad(labels.to(self.device_), (0, 1), value=-100)[:, 1:].reshape(-1).contiguous(), None)
        last_rows = rows if (rows is not None and self.last_layer_rows_only and torch.is_grad_enabled() and not want_hidden and not return_logits
                             and self.dec_layers > 0) else None
        x_before_last, last_args = None, None
        for i in range(self.dec_layers):
            p = f"{lm}layers.{i}."
            la = (self._anchor(p + "mlp.down_proj.weight"), a, p, B, S, self.Hq, self.Hkv, self.D, self.rms_eps, cos, sin, pos, kv_len, krange, kv_lo)
            if i + 1 == self.dec_layers and last_rows is not None:
                x_before_last, last_args = x, la
                x = self._layer(F_.DecoderLayerFn.apply, x, *la, last_rows, ckpt=self.ckpt_plan is None or i < self.ckpt_plan["dec"])
            else:
                x = self._layer(F_.DecoderLayerFn.apply, x, *la, ckpt=self.ckpt_plan is None or i < self.ckpt_plan["dec"])
            if want_hidden and i + 1 < self.dec_layers:
                hidden.append(x.reshape(B, S, -1))
        x = F_.RMSNormFn.apply(x, self._anchor(lm + "norm.weight"), a, lm + "norm.weight", self.rms_eps)
        if want_hidden:
            hidden.append(x.reshape(B, S, -1))                            # ... with the LAST entry taken after the final norm (lm_head(hidden[-1]) == logits)
            hidden = tuple(hidden)
        loss, logits = None, None
        if labels is not None:
            if last_rows is not None:   # x already holds the labelled rows only, in order
                shift, rows = shift.index_select(0, last_rows), None
            if num_items_in_batch is not None:
                denom = torch.as_tensor(num_items_in_batch, device=self.device_, dtype=torch.float32).reshape(1)
            else:
                denom = ops.count_valid(shift)
            loss = F_.LMHeadLossFn.apply(x, self._anchor("lm_head.weight"), a, "lm_head.weight", shift, denom, rows)
            if self._rows_poison is not None:   # label_rows_static promise broken -> NaN loss (see _valid_rows); 0.0 otherwise
                loss, self._rows_poison = loss + self._rows_poison, None
        def _logits(xh=x):
            xs = xh
            if isinstance(logits_to_keep, int) and logits_to_keep > 0:
                xs = xh.reshape(B, S, -1)[:, -logits_to_keep:, :].reshape(-1, xh.shape[-1]).contiguous()
            return F_.LMHeadFn.apply(xs, self._anchor("lm_head.weight"), a, "lm_head.weight").reshape(B, -1, self.V)

        if labels is None or return_logits:
            return AF3Output(loss=loss, logits=_logits(), hidden_states=hidden, audio_hidden_states=audio_hidden)
        # labels given: the reference returns logits beside the loss (modeling_audioflamingo3.py:625-642); here they are built on first access -
        # from the DETACHED final hidden states (no second autograd branch, the step's graph is not kept alive by the output object) and only while
        # the weights are still the ones the loss was computed with: after an optimizer step the lm_head has moved, the logits would no longer
        # belong to `loss`, and the access raises instead (forward(return_logits=True) is the in-graph, forward-time path)
        x_keep, weights_at = (x if last_rows is None else x_before_last).detach(), a.version()

        def _lazy_logits():
            if a.version() != weights_at:
                raise AfkError("output.logits was first read after the parameters changed (optimizer step / load_state_dict): lazy logits would be "
                               "computed with other weights than output.loss; read them before the step or call forward(return_logits=True)")
            with torch.no_grad():
                if last_rows is None:
                    return _logits(x_keep)
                # the step ran its last layer on the labelled rows only: every position's logits need that layer (and the final norm) on every row
                xf = F_.DecoderLayerFn.apply(x_keep, *last_args)
                return _logits(F_.RMSNormFn.apply(xf, self._anchor(lm + "norm.weight"), a, lm + "norm.weight", self.rms_eps))

        return AF3Output(loss=loss, logits_fn=_lazy_logits, hidden_states=hidden, audio_hidden_states=audio_hidden)

    # ------------------------------------------------------------------ generate (greedy; KV cache - SURVEY.md 8(f)-4)
    def _merged_embeddings(self, ids, input_features, input_features_mask):
        """embed_tokens(ids) with the <sound> rows replaced by the projected audio rows (forward() stages a9-a10) -> [B*S, H]"""
        a, lm = self.arena, self._lm
        ids_flat = ids.reshape(-1).contiguous()
        audio, src = None, None
        if input_features is not None:
            audio, n_tok = self.get_audio_features(input_features.to(self.device_), input_features_mask, input_ids=ids)
            src, _ = ops.placeholder_scan(ids_flat, self.audio_token_id)
            if n_tok is not None:
                T3 = audio.shape[0] // input_features.shape[0]
                csum = torch.cumsum(n_tok, 0)
                r = src.clamp_min(0).to(torch.int64)
                win = torch.searchsorted(csum, r, right=True).clamp_max(n_tok.numel() - 1)
                row = win * T3 + (r - (csum - n_tok)[win])
                src = torch.where(src >= 0, row.to(torch.int32), src).contiguous()
        return ops.embed_scatter_fwd(ids_flat, src, a[lm + "embed_tokens.weight"].data, audio)

    cache_headroom = 256  # forward(use_cache=True): positions reserved beyond the prompt; the cache grows by doubling when they run out

    @torch.no_grad()
    def _forward_cached(self, input_ids, inputs_embeds, input_features, input_features_mask, attention_mask, past, logits_to_keep, position_ids=None):
        dev = self.device_
        if inputs_embeds is not None:
            B, n = inputs_embeds.shape[:2]
            x = inputs_embeds.to(dev, torch.bfloat16).reshape(B * n, self.H).contiguous()
            ids = None
        else:
            ids = input_ids.to(dev)
            B, n = ids.shape
        L, nk = self.dec_layers, self.Hkv * self.D
        ref_cache = None
        if AfkKVCache.is_reference_cache(past):
            # the reference's cache object (GenerationMixin hands a DynamicCache to every forward, generation/utils.py:519-640): its tensors are adopted
            # into this implementation's layout for the call and the new positions are appended to it afterwards - the caller keeps ONE cache object
            ref_cache = past
            past = AfkKVCache.adopt(ref_cache, L, self.Hkv, self.D, n, self.cache_headroom, dev, attention_mask)
            if past is not None and attention_mask is not None:
                attention_mask = None    # consumed: the left padding is in past.lo, the new positions are all real
        if past is None:
            lo = torch.zeros(B, device=dev, dtype=torch.int32)
            padded = False
            if attention_mask is not None:
                am = attention_mask.to(dev)
                if not bool(am.all()):
                    lens = am.sum(-1)
                    if not bool((am.flip(-1).cumsum(-1) == torch.minimum(torch.arange(1, n + 1, device=dev)[None], lens[:, None])).all()):
                        raise AfkError("forward(use_cache=True): only LEFT-padded attention_mask is supported (pad the prompt on the left)")
                    lo, padded = (n - lens).to(torch.int32), True
            Smax = n + self.cache_headroom
            Kc = torch.zeros((L, B, Smax, nk), device=dev, dtype=torch.bfloat16)
            Vt = torch.zeros((L, B, self.Hkv, self.D, ops.pad64(Smax)), device=dev, dtype=torch.bfloat16)
            if ids is not None:
                x = self._merged_embeddings(ids, input_features, input_features_mask)
            start = 0
            fast = self.D in (64, 128) and ops.ATTN_IMPL == "lds" and (self.left_pad_on_lds_kernels or not padded)
            kv_lo = lo if padded else None
        else:
            if not isinstance(past, AfkKVCache):
                raise AfkError("forward(past_key_values=...): pass the AfkKVCache a previous forward(use_cache=True) returned, or the reference's DynamicCache")
            if input_features is not None:
                raise AfkError("forward(past_key_values=...): audio belongs to the prefill call (the reference merges it there too)")
            Kc, Vt, lo, start = past.K, past.Vt, past.lo, past.length
            if start + n > Kc.shape[2]:   # out of reserved positions: double the cache
                Smax = max(2 * Kc.shape[2], start + n + self.cache_headroom)
                K2 = torch.zeros((L, B, Smax, nk), device=dev, dtype=torch.bfloat16)
                V2 = torch.zeros((L, B, self.Hkv, self.D, ops.pad64(Smax)), device=dev, dtype=torch.bfloat16)
                K2[:, :, :start].copy_(Kc[:, :, :start])
                V2[..., :start].copy_(Vt[..., :start])
                Kc, Vt = K2, V2
            if ids is not None:
                x = ops.embed_scatter_fwd(ids.reshape(-1).contiguous(), None, self.arena[self._lm + "embed_tokens.weight"].data, None)
            fast, kv_lo = False, None
        ar = torch.arange(start, start + n, device=dev, dtype=torch.int32)
        if position_ids is not None:
            # the rotary positions the caller computed (GenerationMixin: cumsum(attention_mask) - 1 of the whole sequence, sliced to the new tokens)
            pos_rows = position_ids.to(dev).expand(B, n).to(torch.int32).clamp_min(0).reshape(-1).contiguous()
        elif ref_cache is not None:
            # a reference cache and no position_ids: the reference rotates by cache_position = arange(past, past + n) whatever the mask says
            # (Qwen2Model.forward modeling_qwen2.py:361-364) - the keys already in the cache (or the reference's next call on it) follow that convention
            pos_rows = ar[None, :].expand(B, n).reshape(-1).contiguous()
        else:
            pos_rows = (ar[None, :] - lo[:, None]).clamp_min(0).reshape(-1).contiguous()                   # position_ids = cumsum(mask) - 1
        krange = torch.stack([lo[:, None].expand(B, n), torch.maximum(ar[None, :] + 1, lo[:, None])], -1).contiguous()   # [lo, i + 1)
        y = self._decode_layers(x, B, n, start, (Kc, Vt), pos_rows, krange, fast, kv_lo=kv_lo)
        keep = n if not logits_to_keep else min(int(logits_to_keep), n)
        rows = y.reshape(B, n, -1)[:, n - keep:, :].reshape(B * keep, -1).contiguous()
        logits = ops.gemm_nt(rows, self.arena["lm_head.weight"].data).reshape(B, keep, -1)
        new_cache = AfkKVCache(Kc, Vt, lo, start + n)
        if ref_cache is not None:
            return AF3Output(logits=logits, past_key_values=new_cache.write_back(ref_cache, start, n, self.Hkv, self.D))
        return AF3Output(logits=logits, past_key_values=new_cache)

    def _decode_layers(self, x, B, n, start, cache, pos_rows, krange, fast_prefill, start_dev=None, kv_lo=None):
        """all decoder layers on n new positions per sample (rows [B*n, H]) at cache offset `start` (or *start_dev: graph replay).
        cache = (K [L, B, Smax, Hkv*D] post-RoPE keys, Vt [L, B, Hkv, D, Smaxpad] values stored transposed: the layout the interval
        attention kernels read directly).  Attention: prefill on the LDS-staged causal kernel (left-padded prompts: kv_lo); everything else
        (decode steps, other head sizes) on the interval kernel: query row i of sample b sees keys [krange[b,i,0], krange[b,i,1]);
        with start_dev the kernel is given the whole cache length and the interval alone bounds what is visible."""
        a, lm, Hq, Hkv, D = self.arena, self._lm, self.Hq, self.Hkv, self.D
        Kc, Vt = cache
        Smax, Spad = Kc.shape[2], Vt.shape[4]
        nq, nk = Hq * D, Hkv * D
        cos, sin = self._rope_tables(int(self.config.text_config.max_position_embeddings))
        Sk = Smax if start_dev is not None else start + n
        for i in range(self.dec_layers):
            A = lambda k: a[f"{lm}layers.{i}.{k}"]
            h, _ = ops.rmsnorm_fwd(x, A("input_layernorm.weight").data, self.rms_eps)
            qkv = ops.gemm_nt(h, A("self_attn.qkv.weight").data, bias=A("self_attn.qkv.bias").data)
            ops.rope_(qkv, cos, sin, S=n, nheads=Hq + Hkv, D=D, pos=pos_rows)
            ld = qkv.stride(0)
            _lib.call("afk_kv_cache_append", qkv.data_ptr(), ld, nq, Kc[i].data_ptr(), Smax * nk, Vt[i].data_ptr(), Hkv * D * Spad, Spad,
                      ops._p(start_dev), int(start or 0), B, n, Hkv, D, ops._stream())
            if fast_prefill:
                o, _ = ops.attn_fwd(qkv, B, n, Hq, Hkv, D, scale=D ** -0.5, causal=True, kv_lo=kv_lo)
            elif n == 1 and D in (64, 128) and self.decode_splits > 0:
                # one query row per sample: split-KV kernel (the cache is read once, nsplit blocks per head)
                o = torch.empty((B, nq), device=x.device, dtype=torch.bfloat16)
                ns = self.decode_splits
                ws = torch.empty(_lib.load().afk_attn_decode_workspace_floats(B, Hq, D, ns), device=x.device, dtype=torch.float32)
                _lib.call("afk_attn_decode", qkv.data_ptr(), qkv.stride(0), D, Kc[i].data_ptr(), Smax * nk, nk, D, Vt[i].data_ptr(),
                          Hkv * D * Spad, Spad, o.data_ptr(), nq, D, krange.data_ptr(), B, Hq, Hkv, D, float(D ** -0.5), ns,
                          ws.data_ptr(), ops._stream())
            else:
                o = torch.empty((B * n, nq), device=x.device, dtype=torch.bfloat16)
                lse = torch.empty((B, Hq, n), device=x.device, dtype=torch.float32)
                _lib.call("afk_xattn_fwd", qkv.data_ptr(), n * ld, D, ld, Kc[i].data_ptr(), Smax * nk, D, nk, Vt[i].data_ptr(),
                          o.data_ptr(), n * nq, D, nq, lse.data_ptr(), 0, krange.data_ptr(), B, Hq, Hkv, n, Sk, ops.pad64(n), Spad, D,
                          float(D ** -0.5), ops._stream())
            x2 = ops.gemm_nt(o, A("self_attn.o_proj.weight").data, residual=x)
            h2, _ = ops.rmsnorm_fwd(x2, A("post_attention_layernorm.weight").data, self.rms_eps)
            gu = ops.gemm_nt(h2, A("mlp.gate_up.weight").data)
            x = ops.gemm_nt(ops.silu_mul_fwd(gu), A("mlp.down_proj.weight").data, residual=x2)
        y, _ = ops.rmsnorm_fwd(x, a[lm + "norm.weight"].data, self.rms_eps)
        return y

    left_pad_on_lds_kernels = True  # left-padded batches on the LDS-staged causal kernels (kv_lo); False: interval kernels (A/B, tests)
    loss_on_valid_rows_only = True  # lm_head + CE (+ their backward GEMMs) on the rows with a label only (False: all B*S rows, as the reference)
    decode_splits = 8  # key-range splits of the Q = 1 attention (0: use the interval MFMA kernel instead)
    decode_fused_glue = True  # B <= 4: one glue kernel per Linear (csrc/decode_glue.hip) instead of reduce / bias / rope / append / norm / SwiGLU launches

    def _decode_layers_fused(self, x, B, cache, pos_rows, krange, start_dev):
        """one new position per sample, B <= ops.GEMV_MAX_M rows: every Linear = weight-streaming GEMV partials + ONE glue kernel"""
        a, lm, Hq, Hkv, D = self.arena, self._lm, self.Hq, self.Hkv, self.D
        Kc, Vt = cache
        Smax, Spad = Kc.shape[2], Vt.shape[4]
        nq, nk = Hq * D, Hkv * D
        H = x.shape[1]
        dev = x.device
        cos, sin = self._rope_tables(int(self.config.text_config.max_position_embeddings))
        st = ops._stream()
        ns = self.decode_splits
        aws = torch.empty(_lib.load().afk_attn_decode_workspace_floats(B, Hq, D, ns), device=dev, dtype=torch.float32)

        def gemv(inp, w):
            N, K = w.shape
            sp = ops.splitk_plan(B, N, K)
            ws = torch.empty(sp * B * N, device=dev, dtype=torch.float32)
            _lib.call("afk_gemv_partials", inp.data_ptr(), inp.stride(0), w.data_ptr(), w.stride(0), B, N, K, sp, ws.data_ptr(), st)
            return ws, sp

        h, _ = ops.rmsnorm_fwd(x, a[f"{lm}layers.0.input_layernorm.weight"].data, self.rms_eps)
        for i in range(self.dec_layers):
            A = lambda k: a[f"{lm}layers.{i}.{k}"]
            ws, sp = gemv(h, A("self_attn.qkv.weight").data)
            q = torch.empty((B, nq), device=dev, dtype=torch.bfloat16)
            _lib.call("afk_decode_qkv_finish", ws.data_ptr(), sp, B, A("self_attn.qkv.bias").data.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                      pos_rows.data_ptr(), q.data_ptr(), Kc[i].data_ptr(), Smax * nk, Vt[i].data_ptr(), Hkv * D * Spad, Spad, start_dev.data_ptr(),
                      Hq, Hkv, D, st)
            o = torch.empty((B, nq), device=dev, dtype=torch.bfloat16)
            _lib.call("afk_attn_decode", q.data_ptr(), nq, D, Kc[i].data_ptr(), Smax * nk, nk, D, Vt[i].data_ptr(), Hkv * D * Spad, Spad,
                      o.data_ptr(), nq, D, krange.data_ptr(), B, Hq, Hkv, D, float(D ** -0.5), ns, aws.data_ptr(), st)
            ws, sp = gemv(o, A("self_attn.o_proj.weight").data)
            x2, h2 = torch.empty_like(x), torch.empty_like(x)
            _lib.call("afk_decode_residual_rmsnorm", ws.data_ptr(), sp, B, H, x.data_ptr(), A("post_attention_layernorm.weight").data.data_ptr(),
                      float(self.rms_eps), x2.data_ptr(), h2.data_ptr(), st)
            wgu = A("mlp.gate_up.weight").data
            ws, sp = gemv(h2, wgu)
            act = torch.empty((B, wgu.shape[0] // 2), device=dev, dtype=torch.bfloat16)
            _lib.call("afk_decode_swiglu", ws.data_ptr(), sp, B, wgu.shape[0] // 2, act.data_ptr(), st)
            ws, sp = gemv(act, A("mlp.down_proj.weight").data)
            nxt = a[f"{lm}layers.{i + 1}.input_layernorm.weight"] if i + 1 < self.dec_layers else a[lm + "norm.weight"]
            x, h = torch.empty_like(x2), torch.empty_like(x2)
            _lib.call("afk_decode_residual_rmsnorm", ws.data_ptr(), sp, B, H, x2.data_ptr(), nxt.data.data_ptr(), float(self.rms_eps),
                      x.data_ptr(), h.data_ptr(), st)
        return h  # already through the final norm

    decode_chain = os.environ.get("AFK_DECODE_CHAIN", "1") == "1"   # single sequence: one launch per Linear (csrc/decode_chain.hip), five per layer

    def _decode_layers_chain(self, x, cache, pos_rows, krange, start_dev, aws=None, head=None, greedy=None):
        """one new position of ONE sequence: five launches per decoder layer (round 4; ten on the split-K + glue path above).  Every Linear is one
        weight-streaming launch in which a group of waves owns eight complete output rows: the qkv launch normalises the residual stream in its prologue
        and applies bias / RoPE / cache append in its epilogue, o_proj and down_proj add the residual, gate|up normalises in its prologue and multiplies
        silu(gate) * up in its epilogue; the Q = 1 attention merges its key chunks in the last block to finish (afk_attn_decode_fused).  With `head`
        (lm_head weight, rows % 8 == 0) the final RMSNorm + lm_head are one more launch of the same kind and the fp32 logits [1, V] come back; otherwise
        the normalised hidden row."""
        a, lm, Hq, Hkv, D = self.arena, self._lm, self.Hq, self.Hkv, self.D
        Kc, Vt = cache
        Smax, Spad = Kc.shape[2], Vt.shape[4]
        nq, nk = Hq * D, Hkv * D
        H = x.shape[1]
        dev = x.device
        cos, sin = self._rope_tables(int(self.config.text_config.max_position_embeddings))
        st = ops._stream()
        ns = self.decode_splits
        if aws is None:
            aws = self._decode_attn_workspace(dev)
        q = torch.empty((1, nq), device=dev, dtype=torch.bfloat16)
        o = torch.empty((1, nq), device=dev, dtype=torch.bfloat16)
        eps = float(self.rms_eps)
        for i in range(self.dec_layers):
            A = lambda k: a[f"{lm}layers.{i}.{k}"]
            wqkv = A("self_attn.qkv.weight").data
            _lib.call("afk_decode_chain_qkv", x.data_ptr(), A("input_layernorm.weight").data.data_ptr(), eps, wqkv.data_ptr(), wqkv.stride(0), H,
                      A("self_attn.qkv.bias").data.data_ptr(), cos.data_ptr(), sin.data_ptr(), pos_rows.data_ptr(), q.data_ptr(), Kc[i].data_ptr(),
                      Vt[i].data_ptr(), Spad, start_dev.data_ptr(), Hq, Hkv, D, st)
            _lib.call("afk_attn_decode_fused", q.data_ptr(), nq, D, Kc[i].data_ptr(), Smax * nk, nk, D, Vt[i].data_ptr(), Hkv * D * Spad, Spad,
                      o.data_ptr(), nq, D, krange.data_ptr(), 1, Hq, Hkv, D, float(D ** -0.5), ns, aws.data_ptr(), st)
            wo = A("self_attn.o_proj.weight").data
            x2 = torch.empty_like(x)
            _lib.call("afk_decode_chain_linear_residual", o.data_ptr(), wo.data_ptr(), wo.stride(0), H, nq, x.data_ptr(), x2.data_ptr(), st)
            wgu = A("mlp.gate_up.weight").data
            I = wgu.shape[0] // 2
            act = torch.empty((1, I), device=dev, dtype=torch.bfloat16)
            _lib.call("afk_decode_chain_gate_up", x2.data_ptr(), A("post_attention_layernorm.weight").data.data_ptr(), eps, wgu.data_ptr(), wgu.stride(0), I, H,
                      act.data_ptr(), st)
            wd = A("mlp.down_proj.weight").data
            x = torch.empty_like(x2)
            _lib.call("afk_decode_chain_linear_residual", act.data_ptr(), wd.data_ptr(), wd.stride(0), H, I, x2.data_ptr(), x.data_ptr(), st)
        if greedy is not None:   # generate()'s state dict: the lm_head launch leaves (max, argmax) per eight rows, the select launch does everything up to the next step
            g = greedy
            _lib.call("afk_decode_chain_lm_head", x.data_ptr(), a[lm + "norm.weight"].data.data_ptr(), eps, head.data_ptr(), head.stride(0), head.shape[0], H,
                      None, g["part_val"].data_ptr(), g["part_idx"].data_ptr(), st)
            _lib.call("afk_decode_select_greedy", g["part_val"].data_ptr(), g["part_idx"].data_ptr(), g["part_val"].numel(), g["nxt"].data_ptr(),
                      g["tok_buf"].data_ptr(), g["tok_off"], g["state"].data_ptr(), g["emb"].data_ptr(), g["emb"].stride(0), H, g["x0"].data_ptr(), st)
            return None
        if head is not None:
            logits = torch.empty((1, head.shape[0]), device=dev, dtype=torch.float32)
            _lib.call("afk_decode_chain_lm_head", x.data_ptr(), a[lm + "norm.weight"].data.data_ptr(), eps, head.data_ptr(), head.stride(0), head.shape[0], H,
                      logits.data_ptr(), None, None, st)
            return logits
        y, _ = ops.rmsnorm_fwd(x, a[lm + "norm.weight"].data, self.rms_eps)
        return y

    decode_mfma_from = int(os.environ.get("AFK_DECODE_MFMA_FROM", "2"))   # batched decode: sequences per step from which the norm-in-prologue matrix-pipe launches run
    last_layer_rows_only = os.environ.get("AFK_LAST_LAYER_ROWS", "1") == "1"   # training forward with labels: the last decoder layer behind its attention on the labelled rows only
    decode_chain_batch_max = int(os.environ.get("AFK_DECODE_CHAIN_BATCH_MAX", "32"))   # 9 .. 32 sequences: two / four groups of eight through the same launches (8 = the split-K tile path of rounds 3-5 above eight)
    decode_prologue_max = int(os.environ.get("AFK_DECODE_PROLOGUE_MAX", "8"))   # largest batch that takes the RMSNorm in the consumer's prologue (above: a norm launch + plain launches)
    decode_norm_mode = os.environ.get("AFK_DECODE_NORM", "prologue")   # batched decode, four sequences and more: "prologue" | "producer" | "launch" (_decode_layers_chain_batched)
    decode_chain_batch = int(os.environ.get("AFK_DECODE_CHAIN_BATCH", "8"))   # largest batch the one-launch-per-Linear kernels take (0: single sequence only)

    def _decode_layers_chain_batched(self, x, B, cache, pos_rows, krange, start_dev, aws, head):
        """one new position of 2 .. 32 sequences: the single-sequence chain with M input rows per launch (`afk_decode_chain_*_batched`: the weights are still
        read once per step).  Round 6: 2 .. 8 sequences take the RMSNorm in the prologue of the Linear that consumes it (5 launches per layer), 9 .. 32 run as
        groups of eight behind one norm launch per norm (7 per layer); geometries the matrix-pipe launches do not take keep rounds 4-5's form (7 launches per
        layer against ~12 on the split-K + glue path).  -> fp32 logits [B, V]"""
        a, lm, Hq, Hkv, D = self.arena, self._lm, self.Hq, self.Hkv, self.D
        Kc, Vt = cache
        Smax, Spad = Kc.shape[2], Vt.shape[4]
        nq, nk = Hq * D, Hkv * D
        H = x.shape[1]
        dev = x.device
        cos, sin = self._rope_tables(int(self.config.text_config.max_position_embeddings))
        st = ops._stream()
        ns = self.decode_splits
        eps = float(self.rms_eps)
        q = torch.empty((B, nq), device=dev, dtype=torch.bfloat16)
        o = torch.empty((B, nq), device=dev, dtype=torch.bfloat16)
        # round 6: from two sequences on (the matrix-pipe regime of the batched launches) up to eight, no RMSNorm is a launch of its own.  "prologue" (default): the norm in
        # front of qkv / gate|up / lm_head is taken in that Linear's own prologue (afk_decode_chain_*_norm_batched: every block derives the row statistic and
        # normalises the rows it consumes) - 5 launches per layer; "producer": the norm rides behind o_proj / down (afk_decode_chain_linear_residual_norm_batched:
        # the last block to arrive normalises; measured 8 / 5 us of hand-over per launch); "launch": afk_rmsnorm_fwd as in rounds 4-5 (7 launches per layer)
        I0 = a[f"{lm}layers.0.mlp.gate_up.weight"].data.shape[0] // 2
        shapes_ok = B >= self.decode_mfma_from and self._prologue_shapes_ok(head)
        mode = self.decode_norm_mode if shapes_ok else "launch"
        if mode == "prologue" and B > self.decode_prologue_max:
            mode = "launch"   # 9 .. 32 sequences: normalising 16 - 32 rows in EVERY block costs more VALU time than one norm launch (B = 12: 4.20 ms in the prologue, 3.99 with the launch; gate|up at 17 rows: 82 us)
        if mode == "producer":
            cnt = getattr(self, "_chain_norm_counter", None)
            if cnt is None or cnt.device != dev:
                cnt = self._chain_norm_counter = torch.zeros(1, device=dev, dtype=torch.int32)
        h = ssx = None
        ngrp = 1 if B <= 8 else 2 if B <= 16 else 4   # groups of eight sequences the norm-in-prologue launches run (include/afk.h)
        _p = lambda t: None if t is None else t.data_ptr()
        for i in range(self.dec_layers):
            A = lambda k: a[f"{lm}layers.{i}.{k}"]
            wqkv = A("self_attn.qkv.weight").data
            if mode == "prologue":   # ssx: the partial sums of squares the previous layer's down launch left for these rows (None: the first layer takes the statistic itself)
                _lib.call("afk_decode_chain_qkv_norm_batched", x.data_ptr(), x.stride(0), B, A("input_layernorm.weight").data.data_ptr(), eps, wqkv.data_ptr(), wqkv.stride(0), H,
                          A("self_attn.qkv.bias").data.data_ptr(), cos.data_ptr(), sin.data_ptr(), pos_rows.data_ptr(), q.data_ptr(), nq, Kc[i].data_ptr(), Smax * nk,
                          Vt[i].data_ptr(), Hkv * D * Spad, Spad, start_dev.data_ptr(), Hq, Hkv, D, _p(ssx), H // 16, st)
            else:
                if h is None:
                    h, _ = ops.rmsnorm_fwd(x, A("input_layernorm.weight").data, eps)
                _lib.call("afk_decode_chain_qkv_batched", h.data_ptr(), h.stride(0), B, wqkv.data_ptr(), wqkv.stride(0), H, A("self_attn.qkv.bias").data.data_ptr(),
                          cos.data_ptr(), sin.data_ptr(), pos_rows.data_ptr(), q.data_ptr(), nq, Kc[i].data_ptr(), Smax * nk, Vt[i].data_ptr(), Hkv * D * Spad, Spad,
                          start_dev.data_ptr(), Hq, Hkv, D, st)
            _lib.call("afk_attn_decode_fused", q.data_ptr(), nq, D, Kc[i].data_ptr(), Smax * nk, nk, D, Vt[i].data_ptr(), Hkv * D * Spad, Spad,
                      o.data_ptr(), nq, D, krange.data_ptr(), B, Hq, Hkv, D, float(D ** -0.5), ns, aws.data_ptr(), st)
            wo = A("self_attn.o_proj.weight").data
            x2 = torch.empty_like(x)
            h2 = None
            if mode == "producer":
                h2 = torch.empty_like(x)
                _lib.call("afk_decode_chain_linear_residual_norm_batched", o.data_ptr(), nq, B, wo.data_ptr(), wo.stride(0), H, nq, x.data_ptr(), x.stride(0), x2.data_ptr(), H,
                          A("post_attention_layernorm.weight").data.data_ptr(), eps, h2.data_ptr(), H, cnt.data_ptr(), st)
            elif mode == "prologue":
                ss2 = torch.empty((ngrp, 8, H // 16), device=dev, dtype=torch.float32)   # [groups of eight sequences the launch runs: 1, 2 or 4][8][blocks of the launch]
                _lib.call("afk_decode_chain_linear_residual_ss_batched", o.data_ptr(), nq, B, wo.data_ptr(), wo.stride(0), H, nq, x.data_ptr(), x.stride(0), x2.data_ptr(), H,
                          ss2.data_ptr(), st)
            else:
                _lib.call("afk_decode_chain_linear_residual_batched", o.data_ptr(), nq, B, wo.data_ptr(), wo.stride(0), H, nq, x.data_ptr(), x.stride(0), x2.data_ptr(), H, st)
                h2, _ = ops.rmsnorm_fwd(x2, A("post_attention_layernorm.weight").data, eps)
            wgu = A("mlp.gate_up.weight").data
            I = wgu.shape[0] // 2
            act = torch.empty((B, I), device=dev, dtype=torch.bfloat16)
            if mode == "prologue":
                _lib.call("afk_decode_chain_gate_up_norm_batched", x2.data_ptr(), x2.stride(0), B, A("post_attention_layernorm.weight").data.data_ptr(), eps, wgu.data_ptr(),
                          wgu.stride(0), I, H, act.data_ptr(), I, ss2.data_ptr(), H // 16, st)
            else:
                _lib.call("afk_decode_chain_gate_up_batched", h2.data_ptr(), h2.stride(0), B, wgu.data_ptr(), wgu.stride(0), I, H, act.data_ptr(), I, st)
            wd = A("mlp.down_proj.weight").data
            x = torch.empty_like(x2)
            if mode == "producer":   # the norm that follows: the next layer's input_layernorm, or the model's final norm
                nxt = a[f"{lm}layers.{i + 1}.input_layernorm.weight"].data if i + 1 < self.dec_layers else a[lm + "norm.weight"].data
                h = torch.empty_like(x2)
                _lib.call("afk_decode_chain_linear_residual_norm_batched", act.data_ptr(), I, B, wd.data_ptr(), wd.stride(0), H, I, x2.data_ptr(), H, x.data_ptr(), H,
                          nxt.data_ptr(), eps, h.data_ptr(), H, cnt.data_ptr(), st)
            elif mode == "prologue":
                ssx = torch.empty((ngrp, 8, H // 16), device=dev, dtype=torch.float32)
                _lib.call("afk_decode_chain_linear_residual_ss_batched", act.data_ptr(), I, B, wd.data_ptr(), wd.stride(0), H, I, x2.data_ptr(), H, x.data_ptr(), H, ssx.data_ptr(), st)
            else:
                _lib.call("afk_decode_chain_linear_residual_batched", act.data_ptr(), I, B, wd.data_ptr(), wd.stride(0), H, I, x2.data_ptr(), H, x.data_ptr(), H, st)
                h = None
        logits = torch.empty((B, head.shape[0]), device=dev, dtype=torch.float32)
        if mode == "prologue":
            _lib.call("afk_decode_chain_lm_head_norm_batched", x.data_ptr(), x.stride(0), B, a[lm + "norm.weight"].data.data_ptr(), eps, head.data_ptr(), head.stride(0),
                      head.shape[0], H, logits.data_ptr(), head.shape[0], _p(ssx), H // 16, st)
            return logits
        y = h if mode == "producer" else ops.rmsnorm_fwd(x, a[lm + "norm.weight"].data, eps)[0]
        _lib.call("afk_decode_chain_lm_head_batched", y.data_ptr(), y.stride(0), B, head.data_ptr(), head.stride(0), head.shape[0], H, logits.data_ptr(), head.shape[0], st)
        return logits

    def _prologue_shapes_ok(self, head):
        """the norm-in-prologue matrix-pipe launches (afk_decode_chain_*_norm_batched / _linear_residual_ss_batched) take this geometry"""
        H, nq, nk = self.H, self.Hq * self.D, self.Hkv * self.D
        return (H % 64 == 0 and H <= 4096 and nq % 64 == 0 and self.I % 64 == 0 and (self.D // 2) % 16 == 0 and nk % 32 == 0 and head.shape[0] % 32 == 0)

    def _chain_batch_cap(self, head):
        """largest batch of the one-launch-per-Linear decode step: 8 sequences, or (round 6) 32 = four groups of eight where the norm-in-prologue launches apply"""
        cap = min(self.decode_chain_batch, 8)
        if cap == 8 and self.decode_norm_mode == "prologue" and self.decode_chain_batch_max > 8 and self._prologue_shapes_ok(head):
            cap = min(self.decode_chain_batch_max, 32)
        return cap

    def _chain_ok(self, B):
        """single-sequence decode on csrc/decode_chain.hip: one launch per Linear (head sizes of the decode attention kernel, 16-byte rows)"""
        return (self.decode_chain and B == 1 and self.D in (64, 128) and self.decode_splits > 0 and self.H % 8 == 0
                and self.decode_splits * (self.D + 2) <= 4096)

    def _decode_attn_workspace(self, dev):
        """partials + arrival counters of afk_attn_decode_fused for one sequence; the counters start at zero and every launch leaves them at zero"""
        return torch.zeros(_lib.load().afk_attn_decode_workspace_floats(1, self.Hq, self.D, self.decode_splits), device=dev, dtype=torch.float32)

    def _decode_step(self, st):
        """one greedy decode step on static buffers (everything position-dependent lives on the device): HIP-graph capturable"""
        if "x0" in st:   # greedy, one sequence: 5 launches per layer + lm_head + ONE launch for argmax / token / positions / the next embedding row
            self._decode_layers_chain(st["x0"], st["cache"], st["pos1"], st["kr1"], st["cur"], aws=st["aws"], head=st["head"], greedy=st)
            return
        st["nxt"].copy_(self._select_token(self._decode_logits(st), st.get("sampling")))
        (st["advance"] if "advance" in st else st["cur"]).add_(1)   # single sequence: cur, position and the key-range end live in one tensor (generate())

    @torch.no_grad()
    def _beam_search(self, ids, last_hidden, cache, lo, nb, max_new, eos_token_id, pad_token_id, length_penalty, early_stopping):
        """Beam search on the KV cache with GenerationMixin's scoring (transformers/generation/utils.py:3008-3400): every batch row keeps `nb`
        running beams ranked by accumulated log-probability; at each step the best (n_eos + 1) * nb continuations over all beams are drawn, the
        ones among the top nb that end (EOS, or the length limit) compete for the row's nb FINISHED slots with score = sum / generated_length ^
        length_penalty, the best nb that do not end continue.  The loop stops when no running beam can still beat the worst finished one
        (the reference's heuristic: best running sum / current generated length ^ length_penalty; early_stopping = True: as soon as every
        finished slot is filled; "never": the optimistic bound at the maximum length when length_penalty > 0) or the length limit is reached.
        The KV cache is reordered by beam parentage every step (index_select over its batch dimension).  Returns the best finished hypothesis
        per row, padded with pad_token_id (eos if none)."""
        dev = self.device_
        B, S0 = ids.shape
        Kc, Vt = cache
        V = self.V
        eos = [] if eos_token_id is None else ([int(e) for e in eos_token_id] if isinstance(eos_token_id, (list, tuple)) else [int(eos_token_id)])
        keep = (len(eos) + 1) * nb
        rep = torch.arange(B, device=dev).repeat_interleave(nb)
        Kc, Vt = Kc.index_select(1, rep).contiguous(), Vt.index_select(1, rep).contiguous()
        st = {"cache": (Kc, Vt), "lo": lo.index_select(0, rep).contiguous(), "head": self.arena["lm_head.weight"].data,
              "emb": self.arena[self._lm + "embed_tokens.weight"].data, "cur": torch.full((1,), S0, device=dev, dtype=torch.int32),
              "nxt": torch.zeros(B * nb, device=dev, dtype=torch.int64)}
        run_seq = torch.zeros((B, nb, max_new), device=dev, dtype=torch.int64)
        run_score = torch.full((B, nb), -1.0e9, device=dev, dtype=torch.float32)
        run_score[:, 0] = 0.0                                           # all beams of a row start identical: only the first one is live
        fin_seq = torch.zeros((B, nb, max_new), device=dev, dtype=torch.int64)
        fin_len = torch.zeros((B, nb), device=dev, dtype=torch.int64)
        fin_score = torch.full((B, nb), -1.0e9, device=dev, dtype=torch.float32)
        fin_done = torch.zeros((B, nb), device=dev, dtype=torch.bool)
        logits = ops.gemm_nt(last_hidden, st["head"]).float().index_select(0, rep)      # next-token logits after the prompt, one copy per beam
        top_mask = torch.arange(keep, device=dev) < nb
        boff = (torch.arange(B, device=dev) * nb)[:, None]
        can_improve = torch.ones((B, 1), device=dev, dtype=torch.bool)
        for t in range(max_new):
            logp = torch.log_softmax(logits, dim=-1).view(B, nb, V) + run_score[:, :, None]
            cand_score, cand = logp.view(B, nb * V).topk(keep, dim=-1)
            parent, tok = cand // V, cand % V
            cand_seq = run_seq.gather(1, parent[:, :, None].expand(B, keep, max_new)).clone()
            cand_seq[:, :, t] = tok
            ends = torch.zeros_like(tok, dtype=torch.bool)
            for e in eos:
                ends |= tok == e
            if t + 1 == max_new:
                ends[:] = True                                            # the length limit finishes whatever is left
            # finished slots: candidates among the top nb that end, scored sum / generated_length ^ length_penalty
            fscore = cand_score / float((t + 1) ** length_penalty)
            full = fin_done.all(-1, keepdim=True) & (early_stopping is True)
            fscore = fscore + (full | ~can_improve | ~(ends & top_mask[None])).float() * -1.0e9
            ms, mi = torch.cat([fin_score, fscore], 1).topk(nb, dim=-1)
            fin_seq = torch.cat([fin_seq, cand_seq], 1).gather(1, mi[:, :, None].expand(B, nb, max_new))
            fin_len = torch.cat([fin_len, torch.full_like(tok, t + 1)], 1).gather(1, mi)
            fin_done = torch.cat([fin_done, ends & top_mask[None]], 1).gather(1, mi)
            fin_score = ms
            # running beams: the best nb candidates that do not end
            rs, ri = (cand_score + ends.float() * -1.0e9).topk(nb, dim=-1)
            run_seq, run_score = cand_seq.gather(1, ri[:, :, None].expand(B, nb, max_new)), rs
            if t + 1 == max_new:
                break
            # can a running beam still beat the worst finished hypothesis?  (reference heuristic, utils.py:3008-3052)
            hyp_len = (max_new if (early_stopping == "never" and length_penalty > 0.0) else (t + 1))
            best_running = run_score[:, :1] / float(hyp_len ** length_penalty)
            worst_fin = torch.where(fin_done, fin_score.min(-1, keepdim=True).values, torch.full_like(fin_score, -1.0e9))
            can_improve = can_improve & (best_running > worst_fin).any(-1, keepdim=True)
            if not bool(can_improve.any()) or (early_stopping is True and bool(fin_done.all())):
                break
            # advance the model: reorder the cache by parentage, feed the chosen tokens
            src = (parent.gather(1, ri) + boff).reshape(-1)
            Kc.copy_(Kc.index_select(1, src)), Vt.copy_(Vt.index_select(1, src))
            st["nxt"].copy_(tok.gather(1, ri).reshape(-1))
            logits = self._decode_logits(st)
            st["cur"].add_(1)
        best_seq, best_len = fin_seq[:, 0], fin_len[:, 0]
        pad = pad_token_id if pad_token_id is not None else (eos[0] if eos else 0)
        best_seq = torch.where(torch.arange(max_new, device=dev)[None] < best_len[:, None], best_seq, torch.full_like(best_seq, pad))
        return torch.cat([ids, best_seq[:, : int(best_len.max())]], dim=1)

    def _decode_logits(self, st):
        """logits [B, V] (fp32) of the position after st["nxt"]: one pass over the decoder weights, cache append at st["cur"] (not advanced here)"""
        B = st["nxt"].shape[0]
        x = st["emb"].index_select(0, st["nxt"])
        if "pos1" in st:   # single sequence (generate()): views of the step-state tensor, advanced by ONE add per step
            pos1, kr1 = st["pos1"], st["kr1"]
        else:
            pos1 = (st["cur"] - st["lo"]).contiguous()
            kr1 = torch.stack([st["lo"], (st["cur"] + 1).expand(B)], -1).reshape(B, 1, 2).contiguous()
        if (self._chain_ok(1) and 2 <= B <= self._chain_batch_cap(st["head"]) and st["head"].shape[0] % 8 == 0 and self.I % 4 == 0):
            if "aws" not in st:
                st["aws"] = torch.zeros(_lib.load().afk_attn_decode_workspace_floats(B, self.Hq, self.D, self.decode_splits), device=x.device, dtype=torch.float32)
            return self._decode_layers_chain_batched(x.contiguous(), B, st["cache"], pos1, kr1, st["cur"], st["aws"], st["head"])
        if self._chain_ok(B):
            if "aws" not in st:
                st["aws"] = self._decode_attn_workspace(x.device)
            head = st["head"] if st["head"].shape[0] % 8 == 0 else None
            y = self._decode_layers_chain(x.contiguous(), st["cache"], pos1, kr1, st["cur"], aws=st["aws"], head=head)
            if head is not None:
                return y
        elif self.decode_fused_glue and B <= ops.GEMV_MAX_M and self.D in (64, 128) and self.decode_splits > 0 and ops.SPLITK:
            y = self._decode_layers_fused(x.contiguous(), B, st["cache"], pos1, kr1, st["cur"])
        else:
            y = self._decode_layers(x, B, 1, None, st["cache"], pos1, kr1, False, start_dev=st["cur"])
        return ops.gemm_nt(y, st["head"]).float()

    @staticmethod
    def _select_token(logits, sampling=None):
        """greedy argmax, or GenerationMixin's sampling chain on the [B, V] fp32 logits of the new position: temperature -> top-k -> top-p
        (TemperatureLogitsWarper / TopKLogitsWarper / TopPLogitsWarper, transformers/generation/logits_process.py) -> softmax -> multinomial.
        Token selection is O(B*V) index work outside the training hot path: plain torch ops."""
        if not sampling:
            return logits.argmax(-1)
        t, k, p_, gen = sampling["temperature"], sampling["top_k"], sampling["top_p"], sampling["generator"]
        if t and t != 1.0:
            logits = logits / t
        if k and k > 0:
            kth = logits.topk(min(k, logits.shape[-1]), -1).values[..., -1:]
            logits = logits.masked_fill(logits < kth, float("-inf"))
        if p_ is not None and p_ < 1.0:
            sl, si = logits.sort(-1, descending=False)
            drop = sl.softmax(-1).cumsum(-1) <= (1.0 - p_)
            drop[..., -1:] = False  # always keep the most likely token
            logits = logits.masked_fill(drop.scatter(-1, si, drop), float("-inf"))
        return torch.multinomial(logits.softmax(-1), 1, generator=gen).squeeze(-1)

    @torch.no_grad()
    def generate(self, input_ids, input_features=None, input_features_mask=None, attention_mask=None, max_new_tokens=20,
                 do_sample=False, temperature=1.0, top_k=50, top_p=1.0, seed=None, eos_token_id=None, pad_token_id=None, use_cache=True,
                 use_graph=None, num_beams=1, length_penalty=1.0, early_stopping=False, generation_config=None, **kwargs):
        """Greedy decoding, sampling or beam search (GenerationMixin.generate, transformers/generation/utils.py; do_sample with temperature /
        top_k / top_p as its logits warpers apply them; seed -> a device generator, so runs are reproducible; num_beams > 1: beam search with the
        reference's scoring - accumulated log-probabilities, finished hypotheses ranked by sum / length^length_penalty, its early-stop
        heuristic).  generation_config (a transformers.GenerationConfig or anything with the same attributes) supplies defaults for the
        arguments left at theirs, as GenerationMixin merges them.  Cache handling as
        Qwen2Attention.forward modeling_qwen2.py:213-214).  Prefill runs the prompt once and fills a per-layer KV cache; every new
        token then costs one pass over the weights and one Q=1 attention over the cache.  Batches may be LEFT padded
        (attention_mask, as the processor pads): positions count real tokens only and padded keys are never visible.
        The decode step is launch-bound in eager mode (~370 small launches per token), so it is captured once into a HIP graph and
        replayed (use_graph=None: whenever more than 3 tokens are requested)."""
        self._require_hip()
        procs, criteria, streamer = kwargs.pop("logits_processor", None), kwargs.pop("stopping_criteria", None), kwargs.pop("streamer", None)
        for k in ("return_dict_in_generate", "output_scores", "output_logits", "synced_gpus", "use_model_defaults", "tokenizer", "assistant_model"):
            if kwargs.get(k):
                raise AfkError(f"generate({k}=...) is not supported by this implementation")
            kwargs.pop(k, None)
        if kwargs:
            raise AfkError(f"generate(): unknown / unsupported arguments {sorted(kwargs)}")
        hooks = bool(procs) or bool(criteria) or streamer is not None   # GenerationMixin's per-step callbacks: host code between the steps -> eager steps
        gc = generation_config if generation_config is not None else getattr(self, "generation_config", None)
        if gc is not None:  # explicit arguments win; anything still at its default is taken from the generation config
            pick = lambda cur, default, name: getattr(gc, name, None) if (cur == default and getattr(gc, name, None) is not None) else cur
            max_new_tokens = pick(max_new_tokens, 20, "max_new_tokens")
            do_sample, temperature, top_k, top_p = pick(do_sample, False, "do_sample"), pick(temperature, 1.0, "temperature"), pick(top_k, 50, "top_k"), pick(top_p, 1.0, "top_p")
            num_beams, length_penalty, early_stopping = pick(num_beams, 1, "num_beams"), pick(length_penalty, 1.0, "length_penalty"), pick(early_stopping, False, "early_stopping")
            eos_token_id, pad_token_id = pick(eos_token_id, None, "eos_token_id"), pick(pad_token_id, None, "pad_token_id")
        from . import exact as _exact

        if int(max_new_tokens) <= 0:   # GenerationMixin refuses it as well (generation/configuration_utils.py validate())
            raise ValueError(f"`max_new_tokens` must be greater than 0, but is {max_new_tokens}.")
        if _exact.ENABLED and not do_sample and num_beams == 1 and not hooks and type(self).__name__ == "AudioFlamingo3ForConditionalGeneration":
            # AFK_EXACT_FP32=1: greedy decoding by exact-fp32 recomputation of the prefix (exact.py) - the reference's greedy ids with no "confident rows" filter
            return _exact.greedy_generate(self, input_ids, input_features, input_features_mask, attention_mask, max_new_tokens, eos_token_id, pad_token_id)
        if num_beams > 1 and (do_sample or not use_cache):
            raise AfkError("generate(num_beams > 1): beam search is deterministic and runs on the KV cache (no do_sample, no use_cache=False)")
        if hooks and (num_beams > 1 or not use_cache):
            raise AfkError("generate(logits_processor / stopping_criteria / streamer): greedy or sampled decoding on the KV cache only")
        sampling = None
        if do_sample:
            gen = torch.Generator(device=self.device_)
            gen.manual_seed(int(seed) if seed is not None else int(torch.seed() % (2 ** 31)))
            sampling = dict(temperature=float(temperature), top_k=int(top_k or 0), top_p=None if top_p is None else float(top_p), generator=gen)
            use_graph = False  # the sampler draws from a host-side generator object: eager steps
        ids = input_ids.to(self.device_)
        if not use_cache:
            if do_sample:
                raise AfkError("generate(use_cache=False) is the greedy reference path of the tests")
            return self._generate_recompute(ids, input_features, input_features_mask, attention_mask, max_new_tokens, eos_token_id)
        B, S0 = ids.shape
        dev = self.device_
        lo = torch.zeros(B, device=dev, dtype=torch.int32)
        padded = False
        if attention_mask is not None:
            am = attention_mask.to(dev)
            if not bool(am.all()):
                padded = True
                lens = am.sum(-1)
                if not bool((am.flip(-1).cumsum(-1) == torch.minimum(torch.arange(1, S0 + 1, device=dev)[None], lens[:, None])).all()):
                    raise AfkError("generate(): only LEFT-padded attention_mask is supported (pad the prompt on the left)")
                lo = (S0 - lens).to(torch.int32)
        Smax = S0 + max_new_tokens
        L, nk = self.dec_layers, self.Hkv * self.D
        Kc = torch.zeros((L, B, Smax, nk), device=dev, dtype=torch.bfloat16)
        Vt = torch.zeros((L, B, self.Hkv, self.D, ops.pad64(Smax)), device=dev, dtype=torch.bfloat16)
        ar = torch.arange(S0, device=dev, dtype=torch.int32)
        pos_rows = (ar[None, :] - lo[:, None]).clamp_min(0).reshape(-1).contiguous()          # position_ids = cumsum(mask) - 1
        krange = torch.stack([lo[:, None].expand(B, S0), torch.maximum(ar[None, :] + 1, lo[:, None])], -1).contiguous()  # [lo, i+1)
        x = self._merged_embeddings(ids, input_features, input_features_mask)
        fast = self.D in (64, 128) and ops.ATTN_IMPL == "lds" and (self.left_pad_on_lds_kernels or not padded)
        y = self._decode_layers(x, B, S0, 0, (Kc, Vt), pos_rows, krange, fast, kv_lo=lo if padded else None)
        last = y.reshape(B, S0, -1)[:, -1, :].contiguous()
        if num_beams > 1:
            return self._beam_search(ids, last, (Kc, Vt), lo, int(num_beams), int(max_new_tokens), eos_token_id, pad_token_id, float(length_penalty), early_stopping)
        first_logits = ops.gemm_nt(last, self.arena["lm_head.weight"].data).float()
        if procs:
            first_logits = procs(ids, first_logits)
        st = {"cache": (Kc, Vt), "lo": lo, "head": self.arena["lm_head.weight"].data, "emb": self.arena[self._lm + "embed_tokens.weight"].data,
              "cur": torch.full((1,), S0, device=dev, dtype=torch.int32), "sampling": sampling, "nxt": self._select_token(first_logits, sampling)}
        if B == 1:   # one device tensor [lo, key-range end, cache slot, position] -> the views the kernels read; one add per step moves the last three
            state = torch.cat([lo, torch.tensor([S0 + 1, S0], device=dev, dtype=torch.int32), S0 - lo]).contiguous()
            st.update(cur=state[2:3], kr1=state[0:2], pos1=state[3:4], advance=state[1:4], state=state)
            if sampling is None and not hooks and self._chain_ok(1) and st["head"].shape[0] % 8 == 0:   # greedy: token selection and step bookkeeping stay on the device
                tok_buf = torch.zeros(max_new_tokens, device=dev, dtype=torch.int64)
                tok_buf[0] = st["nxt"][0]
                st.update(x0=st["emb"].index_select(0, st["nxt"]).contiguous(), aws=self._decode_attn_workspace(dev), tok_buf=tok_buf, tok_off=1 - S0,
                          part_val=torch.empty(st["head"].shape[0] // 8, device=dev, dtype=torch.float32),
                          part_idx=torch.empty(st["head"].shape[0] // 8, device=dev, dtype=torch.int32))
        if hooks:
            return self._generate_with_hooks(ids, st, first_logits, int(max_new_tokens), procs, criteria, streamer, eos_token_id, pad_token_id)
        on_device = "x0" in st
        n_new = 1
        toks = [st["nxt"].clone()]
        if use_graph is None:
            use_graph = max_new_tokens > 3
        graph = None
        for t in range(1, max_new_tokens):
            if eos_token_id is not None and t % 8 == 0 and bool(((st["tok_buf"][None, :t] if on_device else torch.stack(toks, 1)) == eos_token_id).any(1).all()):
                break
            if use_graph and t == 2:  # step 1 ran eagerly (lazy one-time initialisation inside the library happens outside the capture)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self._decode_step(st)
                graph.replay()  # capture only records: the step for t == 2 itself still has to run
            elif graph is not None:
                graph.replay()
            else:
                self._decode_step(st)
            if not on_device:
                toks.append(st["nxt"].clone())
            n_new = t + 1
        new = st["tok_buf"][None, :n_new].clone() if on_device else torch.stack(toks, 1)
        if eos_token_id is not None:  # everything after a row's first EOS becomes padding (GenerationMixin semantics)
            after = (new == eos_token_id).cumsum(1) - (new == eos_token_id).long() > 0
            new = torch.where(after, torch.full_like(new, pad_token_id if pad_token_id is not None else eos_token_id), new)
        return torch.cat([ids, new], dim=1)

    @torch.no_grad()
    def _generate_with_hooks(self, ids, st, logits, max_new, procs, criteria, streamer, eos_token_id, pad_token_id):
        """GenerationMixin._sample's loop with its per-step callbacks (transformers/generation/utils.py:2730-2830): scores = logits_processor(input_ids,
        logits); token = argmax / multinomial; finished rows emit pad_token_id; streamer.put(tokens) (the prompt first, streamer.end() at the end);
        a row finishes on EOS or when stopping_criteria(input_ids, scores) says so.  Host code runs between the steps, so they are enqueued eagerly
        (no HIP graph, token selection through torch) - the decode kernels are the ones of the graph path."""
        B = ids.shape[0]
        dev = ids.device
        eos = None if eos_token_id is None else torch.as_tensor(eos_token_id, device=dev).reshape(-1)
        pad = pad_token_id if pad_token_id is not None else (int(eos[0]) if eos is not None else 0)
        done = torch.zeros(B, device=dev, dtype=torch.bool)
        seq = ids
        if streamer is not None:
            streamer.put(ids.cpu())
        for t in range(max_new):
            if t > 0:
                logits = self._decode_logits(st)        # appends the K / V of st["nxt"] at the cache slot st["cur"]
                (st["advance"] if "advance" in st else st["cur"]).add_(1)
                if procs:
                    logits = procs(seq, logits)
            tok = self._select_token(logits, st.get("sampling"))
            tok = torch.where(done, torch.full_like(tok, pad), tok)
            seq = torch.cat([seq, tok[:, None]], dim=1)
            if streamer is not None:
                streamer.put(tok.cpu())
            if eos is not None:
                done = done | torch.isin(tok, eos)
            if criteria:
                r = criteria(seq, logits)
                done = done | (r.to(dev).reshape(-1).bool() if torch.is_tensor(r) else torch.full_like(done, bool(r)))
            if bool(done.all()):
                break
            st["nxt"].copy_(tok)
        if streamer is not None:
            streamer.end()
        return seq

    @torch.no_grad()
    def _generate_recompute(self, ids, input_features, input_features_mask, attention_mask, max_new_tokens, eos_token_id):
        """reference path without a cache (every step re-runs the whole prefix through forward()): used by the tests to pin the cache path"""
        if ids.shape[0] != 1 and attention_mask is not None and not bool(attention_mask.all()):
            raise AfkError("the recompute path does not support padded batches")
        for _ in range(max_new_tokens):
            out = self.forward(input_ids=ids, input_features=input_features, input_features_mask=input_features_mask,
                               logits_to_keep=1)
            nxt = out.logits[:, -1, :].float().argmax(-1, keepdim=True)
            ids = torch.cat([ids, nxt], dim=1)
            if eos_token_id is not None and bool((nxt == eos_token_id).all()):
                break
        return ids

    # gradient bookkeeping -------------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = False):
        self.arena.zero_grad()
