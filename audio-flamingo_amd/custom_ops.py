"""The hand-written kernels as REGISTERED PyTorch operators: namespace ``afk`` (``torch.ops.afk.*``), torch.library.custom_op + register_fake +
register_autograd (BASELINE.json north_star: "Python host code on PyTorch-ROCm registers custom ops that call hand-written HIP kernels through a
thin C-ABI layer").

What registration buys over the bare ``torch.autograd.Function`` wrappers (autograd_ops.py, functional.py): the dispatcher knows each op's
schema, its shape function (FakeTensor) and its backward, so ``torch.compile`` / ``torch.export`` trace THROUGH a model built from these ops
(the ctypes call into libafk.so stays an opaque node of the graph instead of a graph break) and ``torch.library.opcheck`` can verify schema,
fake-tensor and autograd registration.  Forward AND backward kernels are registered ops: the backward formulas below are written in terms of
``torch.ops.afk.*_bwd`` so that AOT autograd can trace the backward graph too.

Registered for device type "cuda" (= HIP on ROCm) only: a CPU tensor reaches no kernel and the dispatcher raises - there is no CPU path.

    afk::linear        y = x W^T (+ b)                    nn.Linear (modeling_audioflamingo3.py:109-115, 209-210; modeling_qwen2.py:40-42, 189-192)
    afk::rms_norm      Qwen2RMSNorm                        modeling_qwen2.py:247-252
    afk::layer_norm    nn.LayerNorm                        modeling_audioflamingo3.py:204-208
    afk::attention     SDPA on a fused q|k|v projection    modeling_audioflamingo3.py:174-184, modeling_qwen2.py:195-234 (causal, GQA)
    afk::silu_mul      silu(gate) * up                     modeling_qwen2.py:46-48
    afk::gelu          exact-erf GELU                      activations.py:70-89
    afk::rope          rotate-half RoPE on q|k heads       modeling_qwen2.py:112-135

The AF3 training step itself keeps its layer-level stages (functional.py): they write weight gradients straight into the gradient arena, which
an op returning gradient tensors cannot.  The registered ops are the op-level surface of the same kernels (tests/test_custom_ops_gpu.py runs a
decoder layer built from them under torch.compile and checks it against the layer-level stage).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops

_DEV = "cuda"


def _c(t: Tensor) -> Tensor:
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------------------------- linear
@torch.library.custom_op("afk::linear", mutates_args=(), device_types=_DEV)
def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None) -> Tensor:
    """x [M, K] bf16, weight [N, K] bf16 (nn.Linear layout), bias [N] -> [M, N]   (afk_gemm_nt_bf16, bias fused in the epilogue)"""
    return ops.gemm_nt(_c(x), _c(weight), bias=bias)


@linear.register_fake
def _(x, weight, bias=None):
    return x.new_empty((x.shape[0], weight.shape[0]))


@torch.library.custom_op("afk::linear_bwd", mutates_args=(), device_types=_DEV)
def linear_bwd(dy: Tensor, x: Tensor, weight: Tensor, has_bias: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """dX = dY . W (NN kernel), dW = dY^T . X (TN kernel), db = column sums of dY (empty tensor when has_bias is False)"""
    dy = _c(dy)
    dx = ops.gemm(dy, _c(weight), trans_b=True)
    dw = ops.gemm(dy, _c(x), trans_a=True, trans_b=True)
    db = torch.empty(weight.shape[0] if has_bias else 0, device=dy.device, dtype=dy.dtype)
    if has_bias:
        ops.colsum(dy, db)
    return dx, dw, db


@linear_bwd.register_fake
def _(dy, x, weight, has_bias):
    return x.new_empty(x.shape), weight.new_empty(weight.shape), dy.new_empty((weight.shape[0] if has_bias else 0,))


def _linear_setup(ctx, inputs, output):
    x, weight, bias = inputs
    ctx.save_for_backward(x, weight)
    ctx.has_bias = bias is not None


def _linear_backward(ctx, dy):
    x, weight = ctx.saved_tensors
    dx, dw, db = torch.ops.afk.linear_bwd(dy, x, weight, ctx.has_bias)
    return dx, dw, (db if ctx.has_bias else None)


linear.register_autograd(_linear_backward, setup_context=_linear_setup)


# ---------------------------------------------------------------------------------------------- RMSNorm
@torch.library.custom_op("afk::rms_norm_fwd", mutates_args=(), device_types=_DEV)
def rms_norm_fwd(x: Tensor, weight: Tensor, eps: float) -> Tuple[Tensor, Tensor]:
    return ops.rmsnorm_fwd(_c(x), weight, eps)


@rms_norm_fwd.register_fake
def _(x, weight, eps):
    return x.new_empty(x.shape), x.new_empty((x.numel() // x.shape[-1],), dtype=torch.float32)


@torch.library.custom_op("afk::rms_norm_bwd", mutates_args=(), device_types=_DEV)
def rms_norm_bwd(dy: Tensor, x: Tensor, weight: Tensor, rstd: Tensor) -> Tuple[Tensor, Tensor]:
    dw = torch.empty_like(weight)
    dx = ops.rmsnorm_bwd(_c(x), weight, _c(dy), rstd, dw)
    return dx, dw


@rms_norm_bwd.register_fake
def _(dy, x, weight, rstd):
    return x.new_empty(x.shape), weight.new_empty(weight.shape)


def _rms_setup(ctx, inputs, output):
    x, weight, _ = inputs
    ctx.save_for_backward(x, weight, output[1])


def _rms_backward(ctx, dy, _drstd):
    x, weight, rstd = ctx.saved_tensors
    dx, dw = torch.ops.afk.rms_norm_bwd(dy, x, weight, rstd)
    return dx, dw, None


rms_norm_fwd.register_autograd(_rms_backward, setup_context=_rms_setup)


def rms_norm(x: Tensor, weight: Tensor, eps: float = 1e-6) -> Tensor:
    return torch.ops.afk.rms_norm_fwd(x, weight, eps)[0]


# ---------------------------------------------------------------------------------------------- LayerNorm
@torch.library.custom_op("afk::layer_norm_fwd", mutates_args=(), device_types=_DEV)
def layer_norm_fwd(x: Tensor, weight: Tensor, bias: Tensor, eps: float) -> Tuple[Tensor, Tensor, Tensor]:
    return ops.layernorm_fwd(_c(x), weight, bias, eps)


@layer_norm_fwd.register_fake
def _(x, weight, bias, eps):
    rows = x.numel() // x.shape[-1]
    return x.new_empty(x.shape), x.new_empty((rows,), dtype=torch.float32), x.new_empty((rows,), dtype=torch.float32)


@torch.library.custom_op("afk::layer_norm_bwd", mutates_args=(), device_types=_DEV)
def layer_norm_bwd(dy: Tensor, x: Tensor, weight: Tensor, mean: Tensor, rstd: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    dw, db = torch.empty_like(weight), torch.empty_like(weight)
    dx = ops.layernorm_bwd(_c(x), weight, _c(dy), mean, rstd, dw, db)
    return dx, dw, db


@layer_norm_bwd.register_fake
def _(dy, x, weight, mean, rstd):
    return x.new_empty(x.shape), weight.new_empty(weight.shape), weight.new_empty(weight.shape)


def _ln_setup(ctx, inputs, output):
    x, weight, _, _ = inputs
    ctx.save_for_backward(x, weight, output[1], output[2])


def _ln_backward(ctx, dy, _dm, _dr):
    x, weight, mean, rstd = ctx.saved_tensors
    dx, dw, db = torch.ops.afk.layer_norm_bwd(dy, x, weight, mean, rstd)
    return dx, dw, db, None


layer_norm_fwd.register_autograd(_ln_backward, setup_context=_ln_setup)


def layer_norm(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-5) -> Tensor:
    return torch.ops.afk.layer_norm_fwd(x, weight, bias, eps)[0]


# ---------------------------------------------------------------------------------------------- attention
@torch.library.custom_op("afk::attention_fwd", mutates_args=(), device_types=_DEV)
def attention_fwd(qkv: Tensor, B: int, S: int, Hq: int, Hkv: int, D: int, scale: float, causal: bool) -> Tuple[Tensor, Tensor]:
    """qkv [B*S, (Hq + 2 Hkv) * D] = the fused q|k|v projection output -> o [B*S, Hq*D], lse [B, Hq, pad64(S)]   (LDS-staged kernels, D 64 / 128)"""
    return ops.attn_fwd(_c(qkv), B, S, Hq, Hkv, D, scale=scale, causal=causal)


@attention_fwd.register_fake
def _(qkv, B, S, Hq, Hkv, D, scale, causal):
    return qkv.new_empty((B * S, Hq * D)), qkv.new_empty((B, Hq, ops.pad64(S) if D in (64, 128) else S), dtype=torch.float32)


@torch.library.custom_op("afk::attention_bwd", mutates_args=(), device_types=_DEV)
def attention_bwd(do: Tensor, qkv: Tensor, o: Tensor, lse: Tensor, B: int, S: int, Hq: int, Hkv: int, D: int, scale: float, causal: bool) -> Tensor:
    return ops.attn_bwd(_c(qkv), o, _c(do), lse, B, S, Hq, Hkv, D, scale=scale, causal=causal)


@attention_bwd.register_fake
def _(do, qkv, o, lse, B, S, Hq, Hkv, D, scale, causal):
    return qkv.new_empty(qkv.shape)


def _attn_setup(ctx, inputs, output):
    qkv, B, S, Hq, Hkv, D, scale, causal = inputs
    ctx.save_for_backward(qkv, output[0], output[1])
    ctx.meta = (B, S, Hq, Hkv, D, scale, causal)


def _attn_backward(ctx, do, _dlse):
    qkv, o, lse = ctx.saved_tensors
    return (torch.ops.afk.attention_bwd(do, qkv, o, lse, *ctx.meta),) + (None,) * 7


attention_fwd.register_autograd(_attn_backward, setup_context=_attn_setup)


def attention(qkv: Tensor, B: int, S: int, Hq: int, Hkv: int, D: int, scale: Optional[float] = None, causal: bool = True) -> Tensor:
    return torch.ops.afk.attention_fwd(qkv, B, S, Hq, Hkv, D, float(D ** -0.5 if scale is None else scale), causal)[0]


# ---------------------------------------------------------------------------------------------- SwiGLU / GELU / RoPE
@torch.library.custom_op("afk::silu_mul", mutates_args=(), device_types=_DEV)
def silu_mul(gate_up: Tensor) -> Tensor:
    """[rows, 2I] (gate | up) -> bf16(bf16(silu(gate)) * up)"""
    return ops.silu_mul_fwd(_c(gate_up))


@silu_mul.register_fake
def _(gate_up):
    return gate_up.new_empty((gate_up.shape[0], gate_up.shape[1] // 2))


@torch.library.custom_op("afk::silu_mul_bwd", mutates_args=(), device_types=_DEV)
def silu_mul_bwd(dh: Tensor, gate_up: Tensor) -> Tensor:
    return ops.silu_mul_bwd(_c(gate_up), _c(dh))


@silu_mul_bwd.register_fake
def _(dh, gate_up):
    return gate_up.new_empty(gate_up.shape)


silu_mul.register_autograd(lambda ctx, dh: torch.ops.afk.silu_mul_bwd(dh, ctx.saved_tensors[0]),
                           setup_context=lambda ctx, inputs, output: ctx.save_for_backward(inputs[0]))


@torch.library.custom_op("afk::gelu", mutates_args=(), device_types=_DEV)
def gelu(x: Tensor) -> Tensor:
    return ops.gelu_fwd(_c(x))


@gelu.register_fake
def _(x):
    return x.new_empty(x.shape)


@torch.library.custom_op("afk::gelu_bwd", mutates_args=(), device_types=_DEV)
def gelu_bwd(dy: Tensor, x: Tensor) -> Tensor:
    return ops.gelu_bwd(_c(dy), _c(x))


@gelu_bwd.register_fake
def _(dy, x):
    return x.new_empty(x.shape)


gelu.register_autograd(lambda ctx, dy: torch.ops.afk.gelu_bwd(dy, ctx.saved_tensors[0]),
                       setup_context=lambda ctx, inputs, output: ctx.save_for_backward(inputs[0]))


@torch.library.custom_op("afk::rope", mutates_args=(), device_types=_DEV)
def rope(qkv: Tensor, cos: Tensor, sin: Tensor, S: int, nheads: int, D: int, transpose: bool) -> Tensor:
    """rotate-half RoPE on the first nheads * D columns (the q and k heads) of a fused projection output [B*S, ld]; row r uses position r % S.
    transpose = True applies the transposed rotation (the backward)."""
    out = qkv.clone(memory_format=torch.contiguous_format)
    ops.rope_(out, cos, sin, S=S, nheads=nheads, D=D, backward=transpose)
    return out


@rope.register_fake
def _(qkv, cos, sin, S, nheads, D, transpose):
    return qkv.new_empty(qkv.shape)


def _rope_setup(ctx, inputs, output):
    _, cos, sin, S, nheads, D, transpose = inputs
    ctx.save_for_backward(cos, sin)
    ctx.meta = (S, nheads, D, transpose)


def _rope_backward(ctx, dy):
    cos, sin = ctx.saved_tensors
    S, nheads, D, transpose = ctx.meta
    return (torch.ops.afk.rope(dy, cos, sin, S, nheads, D, not transpose),) + (None,) * 6


rope.register_autograd(_rope_backward, setup_context=_rope_setup)

REGISTERED = ("linear", "linear_bwd", "rms_norm_fwd", "rms_norm_bwd", "layer_norm_fwd", "layer_norm_bwd", "attention_fwd", "attention_bwd",
              "silu_mul", "silu_mul_bwd", "gelu", "gelu_bwd", "rope")


# ---------------------------------------------------------------------------------------------- log-mel frontend and the optimizer launch (round 4)
# The two remaining pieces of the training step outside the stage operators (stage_ops.py): the on-device log-mel (no gradient: raw audio is an input)
# and the fused AdamW launch, whose five buffers are MUTATED in place - declared as such, so a tracer orders it after the backward operators that wrote
# the gradients.  frontend.LogMelFrontend calls afk::logmel.  arena.FusedAdamW keeps the direct C-ABI call: its launches run bucket by bucket INSIDE backward on a
# side stream, and an operator that declares the parameter arena as mutated bumps the version counter every parameter view shares - autograd would refuse the
# backward of the layers still to come, and the arena's W^T-shadow freshness stamps would all go stale.  afk::adamw_step is the operator for hosts that step after backward.
@torch.library.custom_op("afk::logmel", mutates_args=(), device_types=_DEV)
def logmel(wav: Tensor, cosb: Tensor, sinb: Tensor, melT: Tensor, n_mels: int, nbins_pad: int, bf16_out: bool) -> Tensor:
    """WhisperFeatureExtractor's torch path (feature_extraction_whisper.py: STFT 400 / 160, power, mel bank, log10 clamp, per-window max - 8, (x + 4) / 4) in one
    kernel: wav [W, n_samples] fp32 -> [W, n_mels, n_samples / 160] fp32 or bf16"""
    from . import _lib
    from .ops import _stream

    W, n = wav.shape
    T = n // 160
    raw = torch.empty((W, n_mels, T), device=wav.device, dtype=torch.float32)
    wmax = torch.empty(W, device=wav.device, dtype=torch.int32)
    out = raw if not bf16_out else torch.empty((W, n_mels, T), device=wav.device, dtype=torch.bfloat16)
    _lib.call("afk_logmel", wav.data_ptr(), W, n, cosb.data_ptr(), sinb.data_ptr(), nbins_pad, melT.data_ptr(), n_mels, raw.data_ptr(), wmax.data_ptr(),
              out.data_ptr(), int(bf16_out), _stream())
    return out


@logmel.register_fake
def _(wav, cosb, sinb, melT, n_mels, nbins_pad, bf16_out):
    return wav.new_empty((wav.shape[0], n_mels, wav.shape[1] // 160), dtype=torch.bfloat16 if bf16_out else torch.float32)


@torch.library.custom_op("afk::adamw_step", mutates_args=("master", "m", "v", "param"), device_types=_DEV)
def adamw_step(master: Tensor, m: Tensor, v: Tensor, grad: Tensor, param: Tensor, lr: float, beta1: float, beta2: float, eps: float, weight_decay: float,
               step: int, grad_scale: float, max_blocks: int, gate: Optional[Tensor] = None, hyper: Optional[Tensor] = None) -> None:
    """torch.optim.AdamW on a flat range of the arena (fp32 master / m / v, bf16 gradient in, bf16 parameter out: 28 B/param); gate / hyper: device-side launch
    gate and (lr, bias corrections, clip coefficient) for the HIP-graph-replayed step"""
    ops.adamw_step(master, m, v, grad, param, lr=lr, beta1=beta1, beta2=beta2, eps=eps, weight_decay=weight_decay, step=step, grad_scale=grad_scale,
                   max_blocks=max_blocks, gate=gate, hyper=hyper)
