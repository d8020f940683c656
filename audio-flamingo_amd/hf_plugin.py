"""The reference's attention plugin boundary: `AttentionInterface.register(name, fn)`.

`transformers/modeling_utils.py` dispatches every attention module of the reference model through
`ALL_ATTENTION_FUNCTIONS[config._attn_implementation]`; the contract (probed in SURVEY.md Appendix B-2, and the shape
`sdpa_attention_forward` implements, transformers/integrations/sdpa_attention.py:79-166) is

    fn(module, query [B,H,Q,Dh], key [B,Hkv,K,Dh], value [B,Hkv,K,Dh], attention_mask: None | bool [B,1,Q,K],
       dropout: float, scaling: float, **kw)  ->  (out [B,Q,H,Dh] contiguous, attn_weights | None)

`register()` installs the HIP flash-attention kernels behind that contract, forward AND backward (torch.autograd.Function),
so the reference's own `AudioFlamingo3ForConditionalGeneration` on a ROCm device trains and generates through libafk.so with
one call:   audio_flamingo_amd.hf_plugin.register();  model.set_attn_implementation("afk_mi355x").

  * no mask, Q == K, head_dim 64/128     -> LDS-staged kernels (afk_attn2_*), causal iff `module.is_causal`
  * bool mask, Q == K, head_dim 64/128, and the mask is "causal AND key in [lo_b, hi_b)" (left / right padded decoder batches) or
    "every row sees [0, hi_b)" (right-padded encoder windows)
                                         -> the same kernels with kv_lo / kv_len per sample
  * everything else (other interval-shaped masks, Q != K decode steps, other head dims)
                                         -> interval kernels (afk_xattn_*): every query row attends a contiguous key interval
                                            [lo, hi); the mask is converted to those intervals on device and VERIFIED to be
                                            interval-shaped (one host sync per mask tensor OBJECT: the layers of one forward
                                            share the object, a new forward builds a new one)
GQA is native (no repeat_kv).  Dropout must be 0 (the AF3 configs: attention_dropout = 0.0).  bf16 only.  There is no fallback
to torch SDPA: anything unsupported raises.
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch

from . import autograd_ops as A

NAME = "afk_mi355x"
# (weakref to the mask tensor object, its _version, (B, Q, K), intervals).  The key is the tensor OBJECT, never its address: the caching
# allocator hands the next forward's mask the same data_ptr, so an address key would serve a stale batch's intervals.
_mask_cache: list = []
calls = {"lds": 0, "interval": 0}  # which kernel family served the calls (tests assert the plugin really ran)


def _intervals(mask: torch.Tensor) -> torch.Tensor:
    """bool [B,1,Q,K] (True = visible) -> int32 [B,Q,2] key interval per query row; raises if a row is not one interval"""
    m = mask[:, 0]
    K = m.shape[-1]
    cnt = m.sum(-1, dtype=torch.int32)
    lo = m.to(torch.uint8).argmax(-1).to(torch.int32)                      # first visible key (0 when the row is empty)
    last = (K - 1 - m.flip(-1).to(torch.uint8).argmax(-1)).to(torch.int32)
    ok = ((cnt == 0) | (last - lo + 1 == cnt)).all()
    if not bool(ok):
        raise NotImplementedError("afk attention plugin: attention_mask rows must each expose ONE contiguous key interval "
                                  "(causal, bidirectional, left/right padding); got a mask with holes")
    return torch.stack([lo, lo + cnt], dim=-1).contiguous()


def _padding_form(kr: torch.Tensor, Q: int, K: int):
    """Is the interval table one of the two padding forms the LDS-staged kernels take?  -> ("causal" | "full", kv_lo | None, kv_len) or None
         causal:  row i of sample b sees [lo_b, min(i + 1, hi_b))   (left / right padded causal self-attention; rows i < lo_b are padding)
         full:    every row of sample b sees [0, hi_b)              (right-padded bidirectional self-attention)
    Decided on the device, read back once per mask tensor (the caller caches it)."""
    if Q != K:
        return None
    lo_b, hi_b = kr[:, Q - 1, 0], kr[:, Q - 1, 1]        # the last row sees the sample's whole key range in both forms
    lo, end = kr[..., 0], kr[..., 1]
    i1 = torch.arange(1, Q + 1, device=kr.device, dtype=torch.int32)[None]
    live = i1 > lo_b[:, None]                              # query rows that are not left padding
    c_end = torch.maximum(torch.minimum(i1, hi_b[:, None]), lo_b[:, None])
    is_causal = ((~live) | ((lo == lo_b[:, None]) & (end == c_end))).all()
    is_full = ((lo == 0) & (end == hi_b[:, None])).all()
    any_lo = (lo_b > 0).any()
    c, f, a = (bool(x) for x in torch.stack([is_causal, is_full, any_lo]).tolist())
    if c:
        return "causal", (lo_b.contiguous() if a else None), hi_b.contiguous()
    if f:
        return "full", None, hi_b.contiguous()
    return None


def _intervals_of(attention_mask: torch.Tensor, B: int, Q: int, K: int):
    """(intervals, padding form) of the mask tensor the model hands to every layer of ONE forward (same object, same version -> same contents)"""
    shape = (B, Q, K)
    for ref, ver, shp, kr, form in _mask_cache:
        if ref() is attention_mask and ver == attention_mask._version and shp == shape:
            return kr, form
    kr = _intervals(attention_mask[..., :K].expand(B, 1, Q, K))
    form = _padding_form(kr, Q, K)
    _mask_cache[:] = [e for e in _mask_cache if e[0]() is not None][-3:]
    _mask_cache.append((weakref.ref(attention_mask), attention_mask._version, shape, kr, form))
    return kr, form


def _rows(t: torch.Tensor) -> torch.Tensor:
    """[B,H,S,D] (usually a transposed view of the projection output) -> [B*S, H*D] rows; a view when the memory is [B,S,H,D]"""
    B, H, S, D = t.shape
    return t.transpose(1, 2).reshape(B * S, H * D)


def afk_attention(module, query, key, value, attention_mask: Optional[torch.Tensor] = None, dropout: float = 0.0,
                  scaling: Optional[float] = None, is_causal: Optional[bool] = None, **kwargs):
    if query.dtype != torch.bfloat16 or not query.is_cuda:
        raise TypeError(f"afk attention plugin needs bf16 tensors on a ROCm device, got {query.dtype} on {query.device}")
    if dropout and module.training:
        raise NotImplementedError("afk attention plugin: attention dropout is not implemented (AF3 uses 0.0)")
    if kwargs.get("output_attentions"):
        raise NotImplementedError("afk attention plugin: attention weights are never materialised")
    B, Hq, Q, D = query.shape
    Hkv, K = key.shape[1], key.shape[2]
    scale = float(scaling) if scaling is not None else D ** -0.5
    causal = bool(is_causal if is_causal is not None else getattr(module, "is_causal", False)) and Q > 1
    q, k, v = _rows(query), _rows(key), _rows(value)
    krange, form = None, None
    if attention_mask is not None:
        if attention_mask.dtype != torch.bool:
            raise NotImplementedError("afk attention plugin: only boolean masks (the sdpa mask interface) are supported")
        krange, form = _intervals_of(attention_mask, B, Q, K)
    if attention_mask is None and Q == K and D in (64, 128):
        calls["lds"] += 1
        o = A.self_attention(q, k, v, B=B, S=Q, Hq=Hq, Hkv=Hkv, D=D, scale=scale, causal=causal)
    elif form is not None and D in (64, 128):
        # a padded batch whose mask is "causal (or full) AND key in [lo_b, hi_b)": the LDS-staged kernels with kv_lo / kv_len
        calls["lds"] += 1
        o = A.self_attention(q, k, v, B=B, S=Q, Hq=Hq, Hkv=Hkv, D=D, scale=scale, causal=form[0] == "causal", kv_lo=form[1], kv_len=form[2])
    else:
        calls["interval"] += 1
        if attention_mask is not None:
            pass
        elif causal:
            i = torch.arange(Q, device=query.device, dtype=torch.int32) + (K - Q)
            krange = torch.stack([torch.zeros_like(i), i + 1], -1).expand(B, Q, 2).contiguous()
        else:
            krange = None
        o = A.cross_attention(q, k, v, B=B, Sq=Q, Sk=K, H=Hq, D=D, scale=scale, krange=krange, Hkv=Hkv)
    return o.view(B, Q, Hq, D), None


def register(name: str = NAME) -> str:
    """install the plugin in the reference's registries; returns the implementation name to pass to set_attn_implementation()"""
    from transformers import AttentionInterface, AttentionMaskInterface
    from transformers.masking_utils import sdpa_mask

    from . import _lib

    _lib.load()  # fail here, loudly, if libafk.so is missing
    AttentionInterface.register(name, afk_attention)
    AttentionMaskInterface.register(name, sdpa_mask)
    return name
