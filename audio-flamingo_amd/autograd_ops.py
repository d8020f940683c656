"""Stand-alone autograd wrappers over the afk kernels (one op = one Function, gradients returned as tensors).

The AF3 path (functional.py) uses layer-level Functions that write weight gradients straight into the arena; the ops here
are the general-purpose form of the same kernels, used by the Flamingo blocks of BASELINE config 4 (flamingo.py).  The
backward of ``linear`` runs on the transposed-operand MFMA kernels (NN dgrad, TN wgrad): no transposed copies.
"""
from __future__ import annotations

import torch

from . import _lib, ops
from .ops import BF16, _p, _stream


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, residual):
        y = ops.gemm_nt(x, w, bias=b, residual=residual)
        ctx.save_for_backward(x, w)
        ctx.has_b, ctx.has_r = b is not None, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.gemm(dy, w, trans_b=True)                      # dX = dY . W
        dw = ops.gemm(dy, x, trans_a=True, trans_b=True)        # dW = dY^T . X
        db = None
        if ctx.has_b:
            db = torch.empty(w.shape[0], device=dy.device, dtype=BF16)
            ops.colsum(dy, db)
        return dx, dw, db, (dy if ctx.has_r else None)


def linear(x, w, b=None, residual=None):
    """y = x @ w^T (+ b) (+ residual)   x [M, K], w [N, K]"""
    return _Linear.apply(x, w, b, residual)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        y, mean, rstd = ops.layernorm_fwd(x, w, b, eps)
        ctx.save_for_backward(x, w, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        dw, db = torch.empty_like(w), torch.empty_like(w)
        dx = ops.layernorm_bwd(x, w, dy.contiguous(), mean, rstd, dw, db)
        return dx, dw, db, None


def layer_norm(x, w, b, eps=1e-5):
    return _LayerNorm.apply(x, w, b, eps)


class _RMSNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, eps):
        y, rstd = ops.rmsnorm_fwd(x, w, eps)
        ctx.save_for_backward(x, w, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, rstd = ctx.saved_tensors
        dw = torch.empty_like(w)
        dx = ops.rmsnorm_bwd(x, w, dy.contiguous(), rstd, dw)
        return dx, dw, None


def rms_norm(x, w, eps=1e-6):
    return _RMSNorm.apply(x, w, eps)


class _ReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = torch.empty_like(x)
        _lib.call("afk_relu_fwd", x.data_ptr(), y.data_ptr(), x.numel(), _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        _lib.call("afk_relu_bwd", dy.data_ptr(), y.data_ptr(), dx.data_ptr(), dy.numel(), _stream())
        return dx


def relu(x):
    return _ReLU.apply(x)


class _SiluMul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gu):
        ctx.save_for_backward(gu)
        return ops.silu_mul_fwd(gu)

    @staticmethod
    def backward(ctx, dh):
        (gu,) = ctx.saved_tensors
        return ops.silu_mul_bwd(gu, dh.contiguous())


def silu_mul(gu):
    """[rows, 2I] (gate | up) -> silu(gate) * up"""
    return _SiluMul.apply(gu)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add(a, b)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return _Add.apply(a, b)


class _Gate(torch.autograd.Function):
    """y = x + tanh(alpha) * (gate[row] ? h : 0)   (IdeficsGatedCrossAttentionLayer, modeling_idefics.py:792-793,800)"""

    @staticmethod
    def forward(ctx, x, h, alpha, gate):
        rows, D = x.shape
        y = torch.empty_like(x)
        vec = int(alpha.numel() == D and D > 1)
        _lib.call("afk_gate_fwd", x.data_ptr(), h.data_ptr(), alpha.data_ptr(), vec, _p(gate), y.data_ptr(), rows, D, _stream())
        ctx.save_for_backward(h, alpha, gate)
        ctx.vec = vec
        return y

    @staticmethod
    def backward(ctx, dy):
        h, alpha, gate = ctx.saved_tensors
        dy = dy.contiguous()
        rows, D = dy.shape
        dh = torch.empty_like(dy)
        ns = _lib.load().afk_colsum_slices(rows)
        fws = torch.empty((ns + 1) * D, device=dy.device, dtype=torch.float32)
        dalpha = torch.empty_like(alpha)
        _lib.call("afk_gate_bwd", dy.data_ptr(), h.data_ptr(), alpha.data_ptr(), ctx.vec, _p(gate), dh.data_ptr(),
                  fws.data_ptr(), dalpha.data_ptr(), 0, rows, D, _stream())
        return dy, dh, dalpha, None


def gated_residual(x, h, alpha, gate=None):
    return _Gate.apply(x, h, alpha, gate)


class _XAttn(torch.autograd.Function):
    """softmax(q k^T * scale + segment mask) v with Sq != Sk; q [B*Sq, H*D], k / v [B*Sk, H*D]; krange int32 [B, Sq, 2] or None"""

    @staticmethod
    def forward(ctx, q, k, v, krange, B, Sq, Sk, H, D, scale, Hkv=None):
        Hkv = Hkv or H
        o, lse = ops.xattn_fwd(q, k, v, krange, B, Sq, Sk, H, Hkv, D, scale)
        ctx.save_for_backward(q, k, v, o, lse, krange)
        ctx.meta = (B, Sq, Sk, H, D, scale, Hkv)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, krange = ctx.saved_tensors
        B, Sq, Sk, H, D, scale, Hkv = ctx.meta
        do = do.contiguous()
        dev = q.device
        ldo, ldk = H * D, Hkv * D
        dq = torch.empty((B * Sq, ldo), device=dev, dtype=BF16)
        dk = torch.empty((B * Sk, ldk), device=dev, dtype=BF16)
        dv = torch.empty((B * Sk, ldk), device=dev, dtype=BF16)
        ops.xattn_bwd(q, k, v, o, do, lse, krange, B, Sq, Sk, H, Hkv, D, scale, dq, dk, dv)
        return dq, dk, dv, None, None, None, None, None, None, None, None


def cross_attention(q, k, v, *, B, Sq, Sk, H, D, scale, krange=None, Hkv=None):
    """q [B*Sq, H*D], k / v [B*Sk, Hkv*D] (row-strided views allowed); krange int32 [B, Sq, 2] = visible key interval per query"""
    return _XAttn.apply(q, k, v, krange, B, Sq, Sk, H, D, scale, Hkv)


class _SelfAttn(torch.autograd.Function):
    """Sq == Sk self-attention on the LDS-staged kernels (head_dim 64 / 128): full or causal, optional right key padding kv_len[B];
    q [B*S, Hq*D], k / v [B*S, Hkv*D] row-strided views (e.g. slices of a fused projection).  GQA when Hkv < Hq."""

    @staticmethod
    def forward(ctx, q, k, v, kv_len, B, S, Hq, Hkv, D, scale, causal, kv_lo=None):
        spad = ops.pad64(S)
        o = torch.empty((B * S, Hq * D), device=q.device, dtype=BF16)
        lse = torch.zeros((B, Hq, spad), device=q.device, dtype=torch.float32)
        _lib.call("afk_attn2_fwd", q.data_ptr(), S * q.stride(0), D, q.stride(0), k.data_ptr(), S * k.stride(0), D, k.stride(0),
                  v.data_ptr(), S * v.stride(0), D, v.stride(0), o.data_ptr(), S * Hq * D, D, Hq * D, lse.data_ptr(), _p(kv_len),
                  _p(kv_lo), B, Hq, Hkv, S, spad, D, float(scale), int(causal), _stream())
        ctx.save_for_backward(q, k, v, o, lse, kv_len, kv_lo)
        ctx.meta = (B, S, Hq, Hkv, D, scale, causal)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, kv_len, kv_lo = ctx.saved_tensors
        B, S, Hq, Hkv, D, scale, causal = ctx.meta
        do = do.contiguous()
        dev, spad, ldo, ldk = q.device, ops.pad64(S), Hq * D, Hkv * D
        delta = torch.zeros((B, Hq, spad), device=dev, dtype=torch.float32)
        _lib.call("afk_attn2_delta", o.data_ptr(), S * ldo, D, ldo, do.data_ptr(), S * ldo, D, ldo, delta.data_ptr(), B, Hq, S, spad, D, _stream())
        dq = torch.empty((B * S, ldo), device=dev, dtype=BF16)
        dk = torch.empty((B * S, ldk), device=dev, dtype=BF16)
        dv = torch.empty((B * S, ldk), device=dev, dtype=BF16)
        scratch = torch.empty((2, B * S, ldo), device=dev, dtype=BF16) if Hq != Hkv else None
        _lib.call("afk_attn2_bwd", q.data_ptr(), S * q.stride(0), D, q.stride(0), k.data_ptr(), S * k.stride(0), D, k.stride(0),
                  v.data_ptr(), S * v.stride(0), D, v.stride(0), do.data_ptr(), S * ldo, D, ldo, lse.data_ptr(), delta.data_ptr(),
                  dq.data_ptr(), S * ldo, D, ldo, dk.data_ptr(), S * ldk, D, ldk, dv.data_ptr(), S * ldk, D, ldk, _p(kv_len),
                  _p(kv_lo), B, Hq, Hkv, S, spad, D, float(scale), int(causal), _p(scratch), _stream())
        return dq, dk, dv, None, None, None, None, None, None, None, None, None


def self_attention(q, k, v, *, B, S, Hq, Hkv, D, scale, causal, kv_len=None, kv_lo=None):
    """sample b exposes keys [kv_lo[b], kv_len[b]) (int32 [B] on the device, None = no padding on that side); kv_lo needs causal"""
    if D not in (64, 128):
        raise ValueError(f"self_attention: head_dim {D} not supported by the LDS-staged kernels (64 / 128); use cross_attention")
    if kv_lo is not None and not causal:
        raise ValueError("self_attention: kv_lo (left padding) is defined for causal attention only")
    return _SelfAttn.apply(q, k, v, kv_len, B, S, Hq, Hkv, D, scale, causal, kv_lo)


# ---------------------------------------------------------------------------------------------- decoder pieces in general-purpose form
# (the AF3 path keeps its arena-writing layer Functions in functional.py; these return gradients as tensors: config 4's ICL model)
class _Rope(torch.autograd.Function):
    """rotate-half RoPE on the first nheads*D columns of a fused projection [rows, ld] (Qwen2 apply_rotary_pos_emb, modeling_qwen2.py:112-135)"""

    @staticmethod
    def forward(ctx, qkv, cos, sin, S, nheads, D):
        out = qkv.clone()
        ops.rope_(out, cos, sin, S=S, nheads=nheads, D=D)
        ctx.save_for_backward(cos, sin)
        ctx.meta = (S, nheads, D)
        return out

    @staticmethod
    def backward(ctx, dy):
        cos, sin = ctx.saved_tensors
        S, nheads, D = ctx.meta
        dx = dy.clone()
        ops.rope_(dx, cos, sin, S=S, nheads=nheads, D=D, backward=True)
        return dx, None, None, None, None, None


def rope(qkv, cos, sin, *, S, nheads, D):
    return _Rope.apply(qkv, cos, sin, S, nheads, D)


class _FusedSelfAttn(torch.autograd.Function):
    """causal / full self-attention reading q | k | v in place from the fused projection output; backward returns d(q|k|v) as one tensor"""

    @staticmethod
    def forward(ctx, qkv, B, S, Hq, Hkv, D, scale, causal):
        o, lse = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=scale, causal=causal)
        ctx.save_for_backward(qkv, o, lse)
        ctx.meta = (B, S, Hq, Hkv, D, scale, causal)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse = ctx.saved_tensors
        B, S, Hq, Hkv, D, scale, causal = ctx.meta
        return (ops.attn_bwd(qkv, o, do.contiguous(), lse, B, S, Hq, Hkv, D, scale=scale, causal=causal),) + (None,) * 7


def fused_self_attention(qkv, *, B, S, Hq, Hkv, D, scale, causal=True):
    return _FusedSelfAttn.apply(qkv, B, S, Hq, Hkv, D, scale, causal)


class _Embedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, weight):
        ctx.save_for_backward(ids)
        ctx.shape = weight.shape
        return ops.embed_scatter_fwd(ids, None, weight, None)

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        dw = torch.zeros(ctx.shape, device=dout.device, dtype=BF16)
        ops.embed_scatter_bwd(ids, None, dout.contiguous(), dw, None)
        return None, dw


def embedding(ids, weight):
    """weight[ids] with a deterministic fp32-accumulated scatter-add backward; ids int64 [n]"""
    return _Embedding.apply(ids, weight)


class _LMHeadLoss(torch.autograd.Function):
    """lm_head + shifted cross-entropy, chunked, on the rows that carry a label only (functional.LMHeadLossFn in tensor-returning form)"""

    CHUNK = 4096

    @staticmethod
    def forward(ctx, x, w, shift_labels, rows):
        M_all = x.shape[0]
        xs = ops.gather_rows(x, rows) if rows is not None else x
        labs = shift_labels.index_select(0, rows) if rows is not None else shift_labels
        M, V = xs.shape[0], w.shape[0]
        dev = x.device
        denom = ops.count_valid(labs)
        row_loss = torch.empty(M, device=dev, dtype=torch.float32)
        dxs = torch.empty_like(xs)
        dw = torch.empty_like(w)
        chunk = min(_LMHeadLoss.CHUNK, M)
        buf = torch.empty((chunk, V), device=dev, dtype=BF16)
        for i, s in enumerate(range(0, M, chunk)):
            e = min(M, s + chunk)
            logits = buf[: e - s]
            ops.gemm_nt(xs[s:e], w, out=logits)
            ops.ce_fwd_bwd_(logits, labs[s:e], row_loss[s:e], denom, upstream=1.0, write_grad=True)
            ops.gemm(logits, w, out=dxs[s:e], trans_b=True)                                       # dX = dlogits . W
            ops.gemm(logits, xs[s:e], out=dw, trans_a=True, trans_b=True, accumulate=i > 0)       # dW += dlogits^T . X
        loss = torch.empty((), device=dev, dtype=torch.float32)
        ops.loss_reduce(row_loss, denom, loss)
        ctx.save_for_backward(dxs, dw, rows)
        ctx.M_all = M_all
        return loss

    @staticmethod
    def backward(ctx, g):
        dxs, dw, rows = ctx.saved_tensors
        g32 = g.reshape(1).float()
        ops.scale_add_(dxs, dxs, g32, accumulate=False)
        ops.scale_add_(dw, dw, g32, accumulate=False)
        dx = ops.scatter_rows(dxs, rows, ctx.M_all) if rows is not None else dxs
        return dx, dw, None, None


def lm_head_loss(x, w, shift_labels, rows=None):
    """mean CE of (x @ w^T) against shift_labels (-100 = ignore); rows = int64 indices of the labelled rows (None: all)"""
    return _LMHeadLoss.apply(x, w, shift_labels, rows)
