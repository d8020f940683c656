"""Tensor-level wrappers over the C ABI (include/afk.h).

PyTorch is used only as plumbing here: device memory (caching allocator), the current HIP stream, tensor
metadata.  Every function enqueues hand-written gfx950 kernels on ``torch.cuda.current_stream()``.
No function has a CPU / eager fallback: inputs must live on a HIP device (AfkError otherwise).
"""
from __future__ import annotations

import os

import torch

from . import _lib
from ._lib import AfkError

BF16 = torch.bfloat16

GEMM_BIAS, GEMM_GELU, GEMM_RESIDUAL, GEMM_OUT_F32, GEMM_ACCUM, GEMM_SWIGLU_BWD, GEMM_SWIGLU_FWD = 1, 2, 4, 8, 16, 32, 64


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _chk(t: torch.Tensor, dtype=None, name="tensor"):
    if not t.is_cuda:
        raise AfkError(f"{name}: expected a HIP device tensor, got {t.device} (no CPU fallback in this package)")
    if dtype is not None and t.dtype != dtype:
        raise AfkError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t


def pad64(n: int) -> int:
    return (n + 63) // 64 * 64


# ---------------------------------------------------------------------------------------------- GEMM
GEMM_FUSE_ROPE = os.environ.get("AFK_FUSE_ROPE_FWD", "1") != "0"   # rotary embedding inside the qkv GEMM's epilogue (afk_gemm_nt_bf16_rope, round 6)


def gemm_nt_rope(a, b, bias, cos, sin, *, S, rope_cols, D, pos=None):
    """qkv = a @ b^T + bias with rotate-half RoPE on the first rope_cols columns (heads of D columns): one launch (afk_gemm_nt_bf16_rope) when the shape allows -
    head_dim 128, N and rope_cols multiples of 256, enough tiles for the 256 x 256 kernel - else afk_gemm_nt_bf16 + afk_rope_inplace.  The same bits either way
    (tests/test_ops_gpu.py::test_gemm_rope_epilogue_bit_equal)."""
    M, N, K = a.shape[0], b.shape[0], a.shape[1]
    fused = (GEMM_FUSE_ROPE and D == 128 and N % 256 == 0 and rope_cols % 256 == 0 and K % 64 == 0 and ((M + 255) // 256) * (N // 256) >= 192
             and cos.data_ptr() % 16 == 0 and sin.data_ptr() % 16 == 0 and a.stride(1) == 1 and b.stride(1) == 1)
    if fused:
        _chk(a, BF16, "gemm a"), _chk(b, BF16, "gemm b"), _chk(cos, BF16, "rope cos"), _chk(sin, BF16, "rope sin")
        out = torch.empty((M, N), device=a.device, dtype=BF16)
        lanes = pos is None and S % 32 == 0 and cos.shape[0] >= S and sin.shape[0] >= S   # positions = row % S and no 32-row block straddles two samples
        cl, sl = (_rope_lanes(cos, S, D, 1), _rope_lanes(sin, S, D, 1)) if lanes else (None, None)
        _lib.call("afk_gemm_nt_bf16_rope", a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), M, N, K,
                  _p(bias), cos.data_ptr(), sin.data_ptr(), _p(pos), S, rope_cols, _p(cl), _p(sl), _stream())
        return out
    out = gemm_nt(a, b, bias=bias)
    rope_(out, cos, sin, S=S, nheads=rope_cols // D, D=D, pos=pos)
    return out


def gemm_nt(a, b, out=None, *, bias=None, residual=None, res_mod=0, gelu=False, preact_out=None, out_f32=False,
            accumulate=False, alpha=1.0, M=None, N=None, K=None, swiglu_bwd=None, swiglu_fwd_out=None):
    """out[M,N] = epi(alpha * a[M,K] @ b[N,K]^T).  a, b: 2-D bf16 with unit inner stride (row stride free).
    swiglu_fwd_out: b is the fused gate|up weight [2I, K]; out [M, 2I] = gate|up as usual and swiglu_fwd_out [M, I] (contiguous) receives
    bf16(bf16(silu(gate)) * up) from the same launch (AFK_GEMM_SWIGLU_FWD; I % 128 == 0)"""
    _chk(a, BF16, "gemm a"), _chk(b, BF16, "gemm b")
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    M = a.shape[0] if M is None else M
    N = b.shape[0] if N is None else N
    K = a.shape[1] if K is None else K
    assert K <= a.shape[1] and K <= b.shape[1]
    if swiglu_fwd_out is not None:
        assert bias is None and residual is None and not gelu and not accumulate and not out_f32 and preact_out is None and swiglu_bwd is None
        _chk(swiglu_fwd_out, BF16, "gemm swiglu_fwd_out")
        assert N % 256 == 0 and swiglu_fwd_out.is_contiguous() and tuple(swiglu_fwd_out.shape) == (M, N // 2)
        if out is None:
            out = torch.empty((M, N), device=a.device, dtype=BF16)
        _lib.call("afk_gemm_nt_bf16", a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), M, N, K, 0, 0, 0, 0,
                  swiglu_fwd_out.data_ptr(), float(alpha), GEMM_SWIGLU_FWD, _stream())
        return out
    if swiglu_bwd is not None:
        # fused SwiGLU backward epilogue: out [M, 2N] = (dgate | dup), swiglu_bwd = saved gate|up [M, 2N]
        assert bias is None and residual is None and not gelu and not accumulate and not out_f32 and swiglu_bwd.shape[1] == 2 * N
        _chk(swiglu_bwd, BF16, "gemm swiglu_bwd")
        if out is None:
            out = torch.empty((M, 2 * N), device=a.device, dtype=BF16)
        _lib.call("afk_gemm_nt_bf16", a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), M, N, K, 0,
                  swiglu_bwd.data_ptr(), swiglu_bwd.stride(0), 0, 0, float(alpha), GEMM_SWIGLU_BWD, _stream())
        return out
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32 if out_f32 else BF16)
    assert out.dim() == 2 and out.stride(1) == 1 and out.shape[0] >= M and out.shape[1] >= N
    flags = 0
    if bias is not None:
        flags |= GEMM_BIAS
        _chk(bias, BF16, "gemm bias")
    if gelu:
        flags |= GEMM_GELU
    if residual is not None:
        flags |= GEMM_RESIDUAL
        _chk(residual, BF16, "gemm residual")
        assert residual.stride(-1) == 1
    if out.dtype == torch.float32:
        flags |= GEMM_OUT_F32
    if accumulate:
        flags |= GEMM_ACCUM
    if preact_out is not None:
        assert preact_out.stride(0) == out.stride(0) and preact_out.dtype == BF16
    splits = splitk_plan(M, N, K)
    if splits > 1 or M <= GEMV_MAX_M:
        ws = torch.empty(splits * M * N, device=a.device, dtype=torch.float32)
        _lib.call("afk_gemm_nt_bf16_splitk", a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0),
                  M, N, K, _p(bias), _p(residual), residual.stride(0) if residual is not None else 0, res_mod,
                  _p(preact_out), float(alpha), flags, splits, ws.data_ptr(), _stream())
        return out
    _lib.call("afk_gemm_nt_bf16", a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0),
              M, N, K, _p(bias), _p(residual), residual.stride(0) if residual is not None else 0, res_mod,
              _p(preact_out), float(alpha), flags, _stream())
    return out


GEMV_MAX_M = 1  # AFK_GEMV_MAX_M in csrc/gemm.hip
if os.environ.get("AFK_GEMM_NARROW") == "1":  # A/B knob: 8-byte GEMM epilogue
    _lib.call("afk_gemm_set_variant", 16)
SPLITK = os.environ.get("AFK_SPLITK", "1") != "0"


def splitk_plan(M, N, K):
    """number of K splits for an NT GEMM: > 1 only when the output has too few 128x128 tiles to fill the chip (2 x 256 workgroup
    slots) and the reduction is long enough to share - decode-time Linears (M = batch) and weight gradients of narrow layers"""
    if M <= GEMV_MAX_M and SPLITK:  # decode: weight-streaming kernel (csrc/gemm.hip gemv_nt_bf16_kernel), 32 weight rows per workgroup
        blocks = (N + 31) // 32
        return max(1, min(16, (K + 511) // 512, (1536 + blocks - 1) // blocks))
    if not SPLITK or K < 1024:
        return 1
    t256 = ((M + 255) // 256) * ((N + 255) // 256)
    t128 = ((M + 127) // 128) * ((N + 127) // 128)
    if t256 >= 192 or t128 > 300:
        return 1
    return max(1, min(16, K // 256, (512 + t128 - 1) // t128))


def gemm(a, b, out=None, *, trans_a=False, trans_b=False, bias=None, residual=None, res_mod=0, gelu=False, preact_out=None,
         accumulate=False, alpha=1.0, _splits=None):
    """General form of the MFMA GEMM (C ABI afk_gemm_bf16).
        NT: a [M,K],  b [N,K]            (forward)
        NN: a [M,K],  b [K,N]  trans_b   (dgrad: dX = dY . W)
        TN: a [K,M],  b [K,N]  both      (wgrad: dW = dY^T . X; reduction length K free)"""
    _chk(a, BF16, "gemm a"), _chk(b, BF16, "gemm b")
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    if trans_a:
        K, M = a.shape
    else:
        M, K = a.shape
    if trans_b:
        Kb, N = b.shape
    else:
        N, Kb = b.shape
    assert K == Kb, (a.shape, b.shape, trans_a, trans_b)
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=BF16)
    assert out.dim() == 2 and out.stride(1) == 1 and out.shape[0] >= M and out.shape[1] >= N
    flags = 0
    if bias is not None:
        flags |= GEMM_BIAS
    if gelu:
        flags |= GEMM_GELU
    if residual is not None:
        flags |= GEMM_RESIDUAL
    if out.dtype == torch.float32:
        flags |= GEMM_OUT_F32
    if accumulate:
        flags |= GEMM_ACCUM
    if _splits is None and trans_a and trans_b and PEEL_TAIL and bias is None and residual is None and not gelu and preact_out is None:
        plan = peel_plan_256(M, N, K)
        if plan is not None:
            # wave quantisation: T = 8.09 (gate|up wgrad) or 4.05 (down wgrad) rounds of 256 tiles pay for a nearly empty last round.  Peel
            # the last tile rows (or columns) off: the main launch fills whole rounds, the strip runs split-K in one short round.
            axis, cut, s = plan
            kw = dict(trans_a=True, trans_b=True, accumulate=accumulate, alpha=alpha)
            if axis == 0:
                gemm(a[:, :cut], b, out[:cut], _splits=1, **kw)
                gemm(a[:, cut:M], b, out[cut:M], _splits=s, **kw)
            else:
                gemm(a, b[:, :cut], out[:, :cut], _splits=1, **kw)
                gemm(a, b[:, cut:N], out[:, cut:N], _splits=s, **kw)
            return out
    splits = _splits if _splits is not None else (splitk_plan_256(M, N, K) if trans_b else 1)
    if splits > 1:
        ws = torch.empty(splits * M * N, device=a.device, dtype=torch.float32)
        _lib.call("afk_gemm_bf16_splitk", int(trans_a), int(trans_b), a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(),
                  out.stride(0), M, N, K, _p(bias), _p(residual), residual.stride(0) if residual is not None else 0, res_mod,
                  _p(preact_out), float(alpha), flags, splits, ws.data_ptr(), _stream())
        return out
    _lib.call("afk_gemm_bf16", int(trans_a), int(trans_b), a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(),
              out.stride(0), M, N, K, _p(bias), _p(residual), residual.stride(0) if residual is not None else 0, res_mod,
              _p(preact_out), float(alpha), flags, _stream())
    return out


SPLITK_ROUND = int(os.environ.get("AFK_SPLITK_ROUND", "256"))
# Off by default - measured on the full step (same box, 2 runs each): the serial GEMM time falls 314.7 -> 309.8 ms/step (gate|up and down
# weight gradients lose their nearly empty ninth / fifth round), but the default three-stream schedule gets SLOWER, 431.4 -> 433.7 / 437.1 ms:
# its second stream was already filling those tails, and the strip adds two launches and a 60 MB fp32 workspace per weight gradient.
PEEL_TAIL = os.environ.get("AFK_PEEL_TAIL", "0") == "1"


def peel_plan_256(M, N, K):
    """TN GEMM whose 256x256 tiles end in a nearly empty round of 256 workgroups: -> (axis, cut, splits) = run rows (axis 0) / columns
    (axis 1) [0, cut) as the main launch (whole rounds) and the remaining strip with `splits` K-splits in one short round; None = leave it.
    Cost model in rounds: ceil(main tiles / 256) + 1 / splits + 0.08 (second launch + fold) against ceil(tiles / 256)."""
    tm, tn = (M + 255) // 256, (N + 255) // 256
    T = tm * tn
    if T <= 256 or K < 2048:
        return None
    base = (T + 255) // 256
    best = None
    for axis, (t_long, t_other) in enumerate(((tm, tn), (tn, tm))):
        for r in range(1, 9):
            t_tail = r * t_other
            t_main = T - t_tail
            if r >= t_long or t_tail > 128:
                break
            sp = min(16, (K // 64) // 8, SPLITK_ROUND // t_tail)
            if sp < 2:
                continue
            cost = (t_main + 255) // 256 + 1.0 / sp + 0.08
            if cost < base - 0.3 and (best is None or cost < best[0]):
                best = (cost, axis, (t_long - r) * 256, sp)
    return None if best is None else best[1:]


def splitk_plan_256(M, N, K):
    """K splits for the 256x256 transposed-operand kernels (one workgroup per CU): only for outputs with < 128 tiles and K >= 2048"""
    if not SPLITK or K < 2048:
        return 1
    t256 = ((M + 255) // 256) * ((N + 255) // 256)
    if t256 >= 128:
        return 1
    # as many splits as still fit ONE round of 256 workgroups (one per CU): ceil() here gave 275-300 workgroups for the encoder's weight
    # gradients - a second round for 19-44 of them, 55-59 % of the CUs over the launch
    return max(1, min(16, K // 512, SPLITK_ROUND // t256))


COLSUM_FUSED = os.environ.get("AFK_COLSUM_FUSED", "1") == "1"   # one launch (the last row slice folds); 0: partial + fold launches (rounds 1-3)
_COLSUM_COUNTERS = {}


def _colsum_counters(dev, n):
    """arrival counters of afk_colsum_bf16_fused: zero before the first launch, left at zero by every launch; one array per (device, stream) because
    launches on different streams may overlap in time"""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    t = _COLSUM_COUNTERS.get(key)
    if t is None or t.numel() < n:
        t = _COLSUM_COUNTERS[key] = torch.zeros(max(n, 4096), device=dev, dtype=torch.int32)
    return t


def colsum(x, out, *, accumulate=False):
    """out[c] (+)= sum_r x[r][c]   (bias gradient)"""
    rows, cols = x.shape
    ns = _lib.load().afk_colsum_slices(rows)
    ws = torch.empty(ns * cols, device=x.device, dtype=torch.float32)
    if COLSUM_FUSED:
        cnt = _colsum_counters(x.device, (cols + 63) // 64)
        _lib.call("afk_colsum_bf16_fused", x.data_ptr(), x.stride(0), rows, cols, out.data_ptr(), int(accumulate), ws.data_ptr(), cnt.data_ptr(), _stream())
    else:
        _lib.call("afk_colsum_bf16", x.data_ptr(), x.stride(0), rows, cols, out.data_ptr(), int(accumulate), ws.data_ptr(), _stream())
    return out


THIN_BLOCKS = 0  # > 0: cap transposes to that many persistent blocks (set while enqueueing on a side stream beside GEMMs)


def transpose(x, out=None, *, rpad=None):
    """x [R, C] (row stride free) -> out [C, Rpad] with zero-filled tail columns."""
    _chk(x, BF16, "transpose x")
    assert x.dim() == 2 and x.stride(1) == 1
    R, C = x.shape
    rpad = pad64(R) if rpad is None else rpad
    if out is None:
        out = torch.empty((C, rpad), device=x.device, dtype=BF16)
    _lib.call("afk_transpose_bf16", x.data_ptr(), out.data_ptr(), R, C, rpad, x.stride(0), out.stride(0), 1, 1, 0, 0, 0, 0,
              THIN_BLOCKS, _stream())
    return out


def transpose_heads(x, B, S, H, D, ld, spad, out=None):
    """x addressed [b][s][h][d] = base + (b*S+s)*ld + h*D + d  ->  out [B, H, D, spad] (zero padded)."""
    _chk(x, BF16, "transpose_heads x")
    if out is None:
        out = torch.empty((B, H, D, spad), device=x.device, dtype=BF16)
    _lib.call("afk_transpose_bf16", x.data_ptr(), out.data_ptr(), S, D, spad, ld, spad, B, H, S * ld, D, H * D * spad,
              D * spad, 0, _stream())
    return out


# ---------------------------------------------------------------------------------------------- norms
def layernorm_fwd(x, w, b, eps=1e-5):
    _chk(x, BF16, "layernorm x")
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty_like(x)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    _lib.call("afk_layernorm_fwd", x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
              rows, D, float(eps), _stream())
    return y, mean, rstd


def _norm_ws(rows, D, device):
    nb = _lib.load().afk_norm_bwd_blocks(rows)
    return torch.empty(nb * 2 * D, device=device, dtype=torch.float32)


def layernorm_bwd(x, w, dy, mean, rstd, dw, db, *, dx_add=None, accumulate=False, colsum_out=None, colsum_accumulate=False):
    """colsum_out (with dx_add, D % 8 == 0): also colsum_out[c] (+)= sum_r dx[r][c] - the bias gradient of the Linear whose output this norm normalised"""
    D = x.shape[-1]
    rows = x.numel() // D
    dx = torch.empty_like(x)
    if colsum_out is not None:
        assert dx_add is not None and D % 8 == 0 and D <= 4096
        nb = _lib.load().afk_norm_bwd_blocks(rows)
        ws = torch.empty(nb * 3 * D, device=x.device, dtype=torch.float32)
        _lib.call("afk_layernorm_bwd_colsum", x.data_ptr(), w.data_ptr(), dy.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), dx_add.data_ptr(),
                  dw.data_ptr(), db.data_ptr(), int(accumulate), colsum_out.data_ptr(), int(colsum_accumulate), ws.data_ptr(), rows, D, _stream())
        return dx
    ws = _norm_ws(rows, D, x.device)
    _lib.call("afk_layernorm_bwd", x.data_ptr(), w.data_ptr(), dy.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
              _p(dx_add), dw.data_ptr(), db.data_ptr(), int(accumulate), ws.data_ptr(), rows, D, _stream())
    return dx


def rmsnorm_fwd(x, w, eps=1e-6):
    _chk(x, BF16, "rmsnorm x")
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty_like(x)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    _lib.call("afk_rmsnorm_fwd", x.data_ptr(), w.data_ptr(), y.data_ptr(), rstd.data_ptr(), rows, D, float(eps), _stream())
    return y, rstd


def rmsnorm_bwd(x, w, dy, rstd, dw, *, dx_add=None, accumulate=False):
    D = x.shape[-1]
    rows = x.numel() // D
    dx = torch.empty_like(x)
    ws = _norm_ws(rows, D, x.device)
    _lib.call("afk_rmsnorm_bwd", x.data_ptr(), w.data_ptr(), dy.data_ptr(), rstd.data_ptr(), dx.data_ptr(), _p(dx_add),
              dw.data_ptr(), int(accumulate), ws.data_ptr(), rows, D, _stream())
    return dx


# ---------------------------------------------------------------------------------------------- elementwise
def gelu_fwd(x):
    y = torch.empty_like(_chk(x, BF16))
    _lib.call("afk_gelu_fwd", x.data_ptr(), y.data_ptr(), x.numel(), _stream())
    return y


def gelu_bwd(dy, pre, *, colsum_out=None, colsum_accumulate=False):
    """colsum_out: also colsum_out[c] (+)= sum_r dx[r][c] (the bias gradient of the Linear that produced `pre`), by a column-owned form of the same arithmetic"""
    dx = torch.empty_like(_chk(dy, BF16))
    if colsum_out is not None:
        rows, C = dy.shape
        ws = torch.empty(_lib.load().afk_gelu_bwd_colsum_parts(rows) * C, device=dy.device, dtype=torch.float32)
        _lib.call("afk_gelu_bwd_colsum", dy.data_ptr(), pre.data_ptr(), dx.data_ptr(), rows, C, colsum_out.data_ptr(), int(colsum_accumulate), ws.data_ptr(), _stream())
        return dx
    _lib.call("afk_gelu_bwd", dy.data_ptr(), pre.data_ptr(), dx.data_ptr(), dy.numel(), _stream())
    return dx


def silu_mul_fwd(gu):
    rows, I2 = _chk(gu, BF16).shape
    h = torch.empty((rows, I2 // 2), device=gu.device, dtype=BF16)
    _lib.call("afk_silu_mul_fwd", gu.data_ptr(), h.data_ptr(), rows, I2 // 2, _stream())
    return h


def silu_mul_bwd(gu, dh):
    rows, I2 = gu.shape
    dgu = torch.empty_like(gu)
    _lib.call("afk_silu_mul_bwd", gu.data_ptr(), dh.data_ptr(), dgu.data_ptr(), rows, I2 // 2, _stream())
    return dgu


def rope_(buf, cos, sin, *, S, nheads, D, pos=None, backward=False):
    """in-place rotate-half RoPE on the first nheads*D columns of buf [rows, ld]."""
    _chk(buf, BF16, "rope buf")
    rows, ld = buf.shape[0], buf.stride(0)
    _lib.call("afk_rope_inplace", buf.data_ptr(), cos.data_ptr(), sin.data_ptr(), _p(pos), rows, S, ld, nheads, D,
              int(backward), _stream())
    return buf


def rotary_time(x, cos, sin, *, backward=False):
    """Music Flamingo rotary time embedding on encoder rows: x [rows, E] bf16, cos / sin [rows, R] fp32 (R <= E, even)"""
    _chk(x, BF16, "rotary_time x"), _chk(cos, torch.float32, "rotary_time cos"), _chk(sin, torch.float32, "rotary_time sin")
    rows, E = x.shape
    R = cos.shape[-1]
    assert x.is_contiguous() and cos.is_contiguous() and sin.is_contiguous() and cos.numel() == rows * R == sin.numel()
    y = torch.empty_like(x)
    _lib.call("afk_rotary_time", x.data_ptr(), cos.data_ptr(), sin.data_ptr(), y.data_ptr(), rows, E, R, int(backward), _stream())
    return y


def add(a, b, out=None):
    out = torch.empty_like(a) if out is None else out
    _lib.call("afk_add_bf16", a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream())
    return out


def scale_add_(x, y, scale_dev, *, accumulate=False):
    """y = (accumulate ? y : 0) + scale_dev[0] * x  (scale on the device)"""
    _lib.call("afk_scale_add_bf16", x.data_ptr(), y.data_ptr(), x.numel(), _chk(scale_dev, torch.float32).data_ptr(),
              int(accumulate), _stream())
    return y


def cast_f32_bf16(x):
    out = torch.empty(x.shape, device=x.device, dtype=BF16)
    _lib.call("afk_cast_f32_bf16", _chk(x, torch.float32).data_ptr(), out.data_ptr(), x.numel(), _stream())
    return out


def rowsum(xt, C, out, *, accumulate=False):
    """out[r] (+)= sum_{c<C} xt[r][c]"""
    _lib.call("afk_rowsum_bf16", xt.data_ptr(), xt.stride(0), C, out.data_ptr(), xt.shape[0], int(accumulate), _stream())
    return out


# ---------------------------------------------------------------------------------------------- conv stem helpers
def im2col_conv1(x):
    """x [W, C, T] (f32 or bf16, channel-major) -> col [W*T, 3*C] bf16"""
    W, C, T = x.shape
    assert x.is_contiguous() and x.is_cuda
    col = torch.empty((W * T, 3 * C), device=x.device, dtype=BF16)
    _lib.call("afk_im2col_conv1", x.data_ptr(), int(x.dtype == torch.float32), col.data_ptr(), W, C, T, _stream())
    return col


def im2col_conv2(h, W, Tin, C):
    Tout = (Tin - 1) // 2 + 1
    col = torch.empty((W * Tout, 3 * C), device=h.device, dtype=BF16)
    _lib.call("afk_im2col_conv2", h.data_ptr(), col.data_ptr(), W, Tin, Tout, C, _stream())
    return col


def col2im_conv2(dcol, W, Tin, C):
    Tout = (Tin - 1) // 2 + 1
    dh = torch.empty((W * Tin, C), device=dcol.device, dtype=BF16)
    _lib.call("afk_col2im_conv2", dcol.data_ptr(), dh.data_ptr(), W, Tin, Tout, C, _stream())
    return dh


def conv_weight_to_gemm(w, out=None):
    """[Co, Ci, 3] -> [Co, 3*Ci] (tap-major)"""
    Co, Ci, _ = w.shape
    out = torch.empty((Co, 3 * Ci), device=w.device, dtype=BF16) if out is None else out
    _lib.call("afk_conv_weight_permute", w.data_ptr(), out.data_ptr(), Co, Ci, 0, 0, _stream())
    return out


def conv_weight_grad_from_gemm(dwp, dw, *, accumulate=False):
    Co, Ci, _ = dw.shape
    _lib.call("afk_conv_weight_permute", dwp.data_ptr(), dw.data_ptr(), Co, Ci, 1, int(accumulate), _stream())
    return dw


def avgpool2_fwd(x, out_rows, C):
    y = torch.empty((out_rows, C), device=x.device, dtype=BF16)
    _lib.call("afk_avgpool2_fwd", x.data_ptr(), y.data_ptr(), out_rows, C, _stream())
    return y


def avgpool2_bwd(dy, out_rows, C):
    dx = torch.empty((out_rows * 2, C), device=dy.device, dtype=BF16)
    _lib.call("afk_avgpool2_bwd", dy.data_ptr(), dx.data_ptr(), out_rows, C, _stream())
    return dx


# ---------------------------------------------------------------------------------------------- embedding
def placeholder_scan(ids, audio_id):
    ids = _chk(ids.contiguous(), torch.int64, "input_ids")
    n = ids.numel()
    src = torch.empty(n, device=ids.device, dtype=torch.int32)
    cnt = torch.empty(1, device=ids.device, dtype=torch.int32)
    _lib.call("afk_placeholder_scan", ids.data_ptr(), n, int(audio_id), src.data_ptr(), cnt.data_ptr(), _stream())
    return src, cnt


def embed_scatter_fwd(ids, src, embed, audio):
    n, H = ids.numel(), embed.shape[1]
    out = torch.empty((n, H), device=embed.device, dtype=BF16)
    _lib.call("afk_embed_scatter_fwd", ids.data_ptr(), _p(src), embed.data_ptr(), _p(audio), out.data_ptr(), n, H, _stream())
    return out


def embed_scatter_bwd(ids, src, dout, d_embed, d_audio, perm=None):
    """perm: int32 row indices sorted stably by token id (index plumbing: torch.sort) - built here when d_embed is wanted"""
    n, H = ids.numel(), dout.shape[-1]
    if d_embed is not None and perm is None:
        perm = torch.sort(ids.reshape(-1), stable=True).indices.to(torch.int32)
    _lib.call("afk_embed_scatter_bwd", ids.data_ptr(), _p(src), dout.data_ptr(), _p(d_embed), _p(d_audio), _p(perm), n, H, _stream())


# ---------------------------------------------------------------------------------------------- attention
def _row_stat_buffer(B, H, S, spad, device, kernel_writes_tail=True):
    """fp32 [B, H, spad] per-query statistic (lse, delta) of the LDS-staged attention kernels.  The kernels write every query < S, and the
    padding tail [S, spad) of a ragged last tile must read as zero: afk_attn2_fwd (lse) and afk_attn2_bwd_fused (delta) write that tail
    themselves since round 4 - no fill launch at all (the encoder's S = 1500 paid two ATen fills per attention call: 64 of the ~96 per step);
    only the separate delta pass (afk_attn2_delta) leaves it to the host."""
    t = torch.empty((B, H, spad), device=device, dtype=torch.float32)
    if spad > S and not kernel_writes_tail:
        t[:, :, S:].zero_()
    return t


ATTN_PERSIST = os.environ.get("AFK_ATTN_PERSIST", "0") == "1"   # forward on resident blocks + work queue (afk_attn2_fwd_persistent); restrictions in include/afk.h
_ATTN_QUEUES = {}


def _attn_queue(dev):
    """work-queue words of afk_attn2_fwd_persistent: zero before the first launch, left at zero by every launch; one pair per (device, stream)"""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    t = _ATTN_QUEUES.get(key)
    if t is None:
        t = _ATTN_QUEUES[key] = torch.zeros(2, device=dev, dtype=torch.int32)
    return t


ATTN_FUSE_DELTA = os.environ.get("AFK_ATTN_FUSE_DELTA", "1") == "1"   # afk_attn2_bwd_fused (delta inside the dQ kernel) vs delta pass + afk_attn2_bwd
ATTN_IMPL = "lds"  # "lds" = attention_lds.hip (head_dim 64/128), "direct" = attention.hip (also head_dim 32; A/B reference)


def _use_lds(D):
    return ATTN_IMPL == "lds" and D in (64, 128)


def attn_fwd(qkv, B, S, Hq, Hkv, D, *, scale, causal, kv_len=None, kv_lo=None):
    """qkv [B*S, (Hq+2Hkv)*D] fused projection output (q | k | v).  -> o [B*S, Hq*D], lse [B,Hq,Spad]
    kv_len / kv_lo: int32 [B], sample b exposes keys [kv_lo[b], kv_len[b]) (right / left padding; kv_lo needs causal and head_dim 64 / 128)"""
    if kv_lo is not None and not (_use_lds(D) and causal):
        raise _lib.AfkError("attn_fwd: kv_lo (left padding) runs on the LDS-staged causal kernels only; use attn_interval_fwd")
    _chk(qkv, BF16, "qkv")
    ld = qkv.stride(0)
    spad = pad64(S)
    q = qkv
    k = qkv[:, Hq * D:]
    v = qkv[:, (Hq + Hkv) * D:]
    o = torch.empty((B * S, Hq * D), device=qkv.device, dtype=BF16)
    if _use_lds(D):
        lse = _row_stat_buffer(B, Hq, S, spad, qkv.device)
        if ATTN_PERSIST and kv_len is None and kv_lo is None and S % 128 == 0:
            _lib.call("afk_attn2_fwd_persistent", q.data_ptr(), S * ld, D, ld, k.data_ptr(), S * ld, D, ld, v.data_ptr(), S * ld, D, ld,
                      o.data_ptr(), S * Hq * D, D, Hq * D, lse.data_ptr(), None, None, B, Hq, Hkv, S, spad, D, float(scale),
                      int(causal), _attn_queue(qkv.device).data_ptr(), _stream())
            return o, lse
        _lib.call("afk_attn2_fwd", q.data_ptr(), S * ld, D, ld, k.data_ptr(), S * ld, D, ld, v.data_ptr(), S * ld, D, ld,
                  o.data_ptr(), S * Hq * D, D, Hq * D, lse.data_ptr(), _p(kv_len), _p(kv_lo), B, Hq, Hkv, S, spad, D, float(scale),
                  int(causal), _stream())
        return o, lse
    vt = transpose_heads(v, B, S, Hkv, D, ld, spad)
    lse = torch.empty((B, Hq, S), device=qkv.device, dtype=torch.float32)
    _lib.call("afk_attn_fwd", q.data_ptr(), S * ld, D, ld, k.data_ptr(), S * ld, D, ld, vt.data_ptr(), o.data_ptr(),
              S * Hq * D, D, Hq * D, lse.data_ptr(), _p(kv_len), B, Hq, Hkv, S, spad, D, float(scale), int(causal), _stream())
    return o, lse


ATTN_FUSE_ROPE_BWD = os.environ.get("AFK_FUSE_ROPE_BWD", "1") != "0"   # rotary backward inside the attention backward (afk_attn2_bwd_fused_rope, round 6)


_ROPE_LANES = {}   # (table data_ptr, version, rows, D) -> lane-major copy (afk_rope_lanes_table); a handful of entries (one per table the model caches)


def _rope_lanes(table, S, D, form=0):
    """lane-major copy of a rotary table for the dQ epilogue of afk_attn2_bwd_fused_rope, cached per table tensor (weakly: a new tensor at the same address gets
    a fresh copy because its version / shape is part of the key and the entry holds a weak reference to the source)"""
    import weakref

    key = (table.data_ptr(), table._version, int(table.shape[0]), D, form)
    hit = _ROPE_LANES.get(key)
    if hit is not None and hit[0]() is table:
        return hit[1]
    rows = int(table.shape[0])
    out = torch.empty(((rows + 31) // 32) * 32 * D, device=table.device, dtype=BF16)
    _lib.call("afk_rope_lanes_table", table.data_ptr(), out.data_ptr(), rows, D, form, _stream())
    if len(_ROPE_LANES) > 64:
        _ROPE_LANES.clear()
    _ROPE_LANES[key] = (weakref.ref(table), out)
    return out


def _attn_bwd_impl(qkv, o, do, lse, B, S, Hq, Hkv, D, *, scale, causal, kv_len=None, kv_lo=None, rope=None):
    """-> (dqkv, rotated): rotated = the rotary backward has been applied inside the kernels"""
    if kv_lo is not None and not (_use_lds(D) and causal and lse.shape[-1] == pad64(S)):
        raise _lib.AfkError("attn_bwd: kv_lo (left padding) runs on the LDS-staged causal kernels only; use attn_interval_bwd")
    ld = qkv.stride(0)
    spad = pad64(S)
    q = qkv
    k = qkv[:, Hq * D:]
    v = qkv[:, (Hq + Hkv) * D:]
    dev = qkv.device
    ldo = Hq * D
    dqkv = torch.empty_like(qkv)
    ldd = dqkv.stride(0)
    dq = dqkv
    dk = dqkv[:, Hq * D:]
    dv = dqkv[:, (Hq + Hkv) * D:]
    if _use_lds(D) and lse.shape[-1] == spad:
        delta = _row_stat_buffer(B, Hq, S, spad, dev, kernel_writes_tail=ATTN_FUSE_DELTA)
        scratch = torch.empty((2, B * S, Hq * D), device=dev, dtype=BF16) if Hq != Hkv else None
        if ATTN_FUSE_DELTA and rope is not None and ATTN_FUSE_ROPE_BWD and rope[0].data_ptr() % 16 == 0 and rope[1].data_ptr() % 16 == 0:
            cos, sin, pos = rope
            _chk(cos, BF16, "rope cos"), _chk(sin, BF16, "rope sin")
            cl, sl = _rope_lanes(cos, S, D) if pos is None and cos.shape[0] >= S else None, _rope_lanes(sin, S, D) if pos is None and sin.shape[0] >= S else None
            _lib.call("afk_attn2_bwd_fused_rope", q.data_ptr(), S * ld, D, ld, k.data_ptr(), S * ld, D, ld, v.data_ptr(), S * ld, D, ld,
                      o.data_ptr(), S * ldo, D, ldo, do.data_ptr(), S * ldo, D, ldo, lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), S * ldd, D, ldd,
                      dk.data_ptr(), S * ldd, D, ldd, dv.data_ptr(), S * ldd, D, ldd, _p(kv_len), _p(kv_lo), B, Hq, Hkv, S, spad, D,
                      float(scale), int(causal), _p(scratch), cos.data_ptr(), sin.data_ptr(), _p(pos), _p(cl), _p(sl), _stream())
            return dqkv, True
        if ATTN_FUSE_DELTA:   # delta = rowsum(dO o O) inside the dQ kernel, which runs ahead of the dK/dV sweep: one pass over O and dO less
            _lib.call("afk_attn2_bwd_fused", q.data_ptr(), S * ld, D, ld, k.data_ptr(), S * ld, D, ld, v.data_ptr(), S * ld, D, ld,
                      o.data_ptr(), S * ldo, D, ldo, do.data_ptr(), S * ldo, D, ldo, lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), S * ldd, D, ldd,
                      dk.data_ptr(), S * ldd, D, ldd, dv.data_ptr(), S * ldd, D, ldd, _p(kv_len), _p(kv_lo), B, Hq, Hkv, S, spad, D,
                      float(scale), int(causal), _p(scratch), _stream())
            return dqkv, False
        _lib.call("afk_attn2_delta", o.data_ptr(), S * ldo, D, ldo, do.data_ptr(), S * ldo, D, ldo, delta.data_ptr(), B, Hq, S,
                  spad, D, _stream())
        _lib.call("afk_attn2_bwd", q.data_ptr(), S * ld, D, ld, k.data_ptr(), S * ld, D, ld, v.data_ptr(), S * ld, D, ld,
                  do.data_ptr(), S * ldo, D, ldo, lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), S * ldd, D, ldd,
                  dk.data_ptr(), S * ldd, D, ldd, dv.data_ptr(), S * ldd, D, ldd, _p(kv_len), _p(kv_lo), B, Hq, Hkv, S, spad, D,
                  float(scale), int(causal), _p(scratch), _stream())
        return dqkv, False
    delta = torch.empty((B, Hq, S), device=dev, dtype=torch.float32)
    _lib.call("afk_attn_delta", o.data_ptr(), S * ldo, D, ldo, do.data_ptr(), S * ldo, D, ldo, delta.data_ptr(), B, Hq, S, D,
              _stream())
    qt = transpose_heads(q, B, S, Hq, D, ld, spad)
    kt = transpose_heads(k, B, S, Hkv, D, ld, spad)
    dot = transpose_heads(do, B, S, Hq, D, ldo, spad)
    _lib.call("afk_attn_bwd", q.data_ptr(), S * ld, D, ld, k.data_ptr(), S * ld, D, ld, v.data_ptr(), S * ld, D, ld,
              do.data_ptr(), S * ldo, D, ldo, qt.data_ptr(), kt.data_ptr(), dot.data_ptr(), lse.data_ptr(), delta.data_ptr(),
              dq.data_ptr(), S * ldd, D, ldd, dk.data_ptr(), S * ldd, D, ldd, dv.data_ptr(), S * ldd, D, ldd, _p(kv_len),
              B, Hq, Hkv, S, spad, D, float(scale), int(causal), _stream())
    return dqkv, False


def attn_bwd(qkv, o, do, lse, B, S, Hq, Hkv, D, *, scale, causal, kv_len=None, kv_lo=None, rope=None):
    """-> dqkv [B*S, (Hq+2Hkv)*D].  rope = (cos, sin, pos or None): the gradient of the rotary embedding is applied to the q | k columns as well - inside the
    attention backward kernels where the path allows (afk_attn2_bwd_fused_rope: LDS kernels with the fused delta), by afk_rope_inplace(backward) otherwise:
    the same bits either way (tests/test_ops_gpu.py::test_attention_backward_fused_rope_bit_equal)."""
    dqkv, rotated = _attn_bwd_impl(qkv, o, do, lse, B, S, Hq, Hkv, D, scale=scale, causal=causal, kv_len=kv_len, kv_lo=kv_lo, rope=rope)
    if rope is not None and not rotated:
        rope_(dqkv, rope[0], rope[1], S=S, nheads=Hq + Hkv, D=D, pos=rope[2], backward=True)
    return dqkv


def xattn_fwd(q, k, v, krange, B, Sq, Sk, Hq, Hkv, D, scale):
    """interval attention (afk_xattn_fwd): query row i of sample b sees keys [krange[b,i,0], krange[b,i,1]) (None: all Sk keys; an empty
    interval yields a zero row).  q [B*Sq, >=Hq*D], k / v [B*Sk, >=Hkv*D] row-strided views.  -> o [B*Sq, Hq*D], lse [B, Hq, Sq]"""
    sqp, skp = pad64(Sq), pad64(Sk)
    vt = transpose_heads(v, B, Sk, Hkv, D, v.stride(0), skp)
    o = torch.empty((B * Sq, Hq * D), device=q.device, dtype=BF16)
    lse = torch.empty((B, Hq, Sq), device=q.device, dtype=torch.float32)
    _lib.call("afk_xattn_fwd", q.data_ptr(), Sq * q.stride(0), D, q.stride(0), k.data_ptr(), Sk * k.stride(0), D, k.stride(0),
              vt.data_ptr(), o.data_ptr(), Sq * Hq * D, D, Hq * D, lse.data_ptr(), 0, _p(krange), B, Hq, Hkv, Sq, Sk, sqp, skp, D,
              float(scale), _stream())
    return o, lse


def xattn_bwd(q, k, v, o, do, lse, krange, B, Sq, Sk, Hq, Hkv, D, scale, dq, dk, dv):
    """backward of xattn_fwd; dq [B*Sq, >=Hq*D], dk / dv [B*Sk, >=Hkv*D] are written in place (row-strided views allowed)"""
    dev = q.device
    sqp, skp = pad64(Sq), pad64(Sk)
    ldo = Hq * D
    delta = torch.empty((B, Hq, Sq), device=dev, dtype=torch.float32)
    _lib.call("afk_attn_delta", o.data_ptr(), Sq * ldo, D, ldo, do.data_ptr(), Sq * ldo, D, ldo, delta.data_ptr(), B, Hq, Sq, D, _stream())
    qt = transpose_heads(q, B, Sq, Hq, D, q.stride(0), sqp)
    kt = transpose_heads(k, B, Sk, Hkv, D, k.stride(0), skp)
    dot = transpose_heads(do, B, Sq, Hq, D, ldo, sqp)
    _lib.call("afk_xattn_bwd", q.data_ptr(), Sq * q.stride(0), D, q.stride(0), k.data_ptr(), Sk * k.stride(0), D, k.stride(0),
              v.data_ptr(), Sk * v.stride(0), D, v.stride(0), do.data_ptr(), Sq * ldo, D, ldo, qt.data_ptr(), kt.data_ptr(),
              dot.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), Sq * dq.stride(0), D, dq.stride(0),
              dk.data_ptr(), Sk * dk.stride(0), D, dk.stride(0), dv.data_ptr(), Sk * dv.stride(0), D, dv.stride(0), 0, _p(krange),
              B, Hq, Hkv, Sq, Sk, sqp, skp, D, float(scale), _stream())


def attn_interval_fwd(qkv, krange, B, S, Hq, Hkv, D, *, scale):
    """self-attention on a fused q|k|v projection with a per-query key interval (left / right / both-side padding, causal or not)"""
    _chk(qkv, BF16, "qkv")
    return xattn_fwd(qkv, qkv[:, Hq * D:], qkv[:, (Hq + Hkv) * D:], krange, B, S, S, Hq, Hkv, D, scale)


def attn_interval_bwd(qkv, o, do, lse, krange, B, S, Hq, Hkv, D, *, scale):
    dqkv = torch.empty_like(qkv)
    xattn_bwd(qkv, qkv[:, Hq * D:], qkv[:, (Hq + Hkv) * D:], o, do, lse, krange, B, S, S, Hq, Hkv, D, scale,
              dqkv, dqkv[:, Hq * D:], dqkv[:, (Hq + Hkv) * D:])
    return dqkv


# ---------------------------------------------------------------------------------------------- row gather / scatter
def gather_rows(x, rows):
    """out[i] = x[rows[i]]   (x [M, H] contiguous bf16, rows int64 on the device): the embedding-gather kernel on an activation matrix"""
    _chk(x, BF16, "gather_rows x")
    assert x.is_contiguous() and rows.dtype == torch.int64
    return embed_scatter_fwd(rows, None, x, None)


def scatter_rows(src, rows, M):
    """out [M, H] zeros except out[rows[i]] = src[i]   (rows unique, ascending)"""
    out = torch.zeros((M, src.shape[1]), device=src.device, dtype=BF16)
    perm = torch.arange(rows.numel(), device=src.device, dtype=torch.int32)
    embed_scatter_bwd(rows, None, src, out, None, perm=perm)
    return out


# ---------------------------------------------------------------------------------------------- loss
def count_valid(labels):
    out = torch.empty(1, device=labels.device, dtype=torch.float32)
    _lib.call("afk_count_valid", labels.data_ptr(), labels.numel(), out.data_ptr(), _stream())
    return out


def ce_fwd_bwd_(logits, shift_labels, row_loss, denom, *, upstream=1.0, write_grad=True):
    rows, V = logits.shape
    _lib.call("afk_ce_fwd_bwd", logits.data_ptr(), logits.stride(0), rows, V, shift_labels.data_ptr(), row_loss.data_ptr(),
              denom.data_ptr(), float(upstream), int(write_grad), _stream())


def loss_reduce(row_loss, denom, loss, *, accumulate=False):
    _lib.call("afk_loss_reduce", row_loss.data_ptr(), row_loss.numel(), denom.data_ptr(), loss.data_ptr(), int(accumulate),
              _stream())
    return loss


# ---------------------------------------------------------------------------------------------- optimizer
def adamw_step(master, m, v, grad, param, *, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, max_blocks=0, gate=None, hyper=None):
    """gate: optional device int32[1]; the launch leaves every buffer untouched when it reads 0.
    hyper: optional device float32[4] = (lr, 1 - beta1^t, sqrt(1 - beta2^t), gradient multiplier) overriding lr / step (HIP-graph replay);
    hyper[3] multiplies every gradient (global-norm clip coefficient, 1 = off)"""
    _lib.call("afk_adamw_step", master.data_ptr(), m.data_ptr(), v.data_ptr(), grad.data_ptr(), param.data_ptr(), param.numel(),
              float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step), float(grad_scale),
              int(max_blocks), _p(gate), _p(hyper), _stream())


def adamw_step_t(master, m, v, grad, param, shadow, N, K, *, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, max_blocks=0, gate=None, hyper=None):
    """afk_adamw_step on one 2-D weight [N, K] (flat views of the arena) that also writes shadow [K, ld] = the K-major copy of the updated weight"""
    _lib.call("afk_adamw_step_t", master.data_ptr(), m.data_ptr(), v.data_ptr(), grad.data_ptr(), param.data_ptr(), shadow.data_ptr(), int(N), int(K),
              shadow.stride(0), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step), float(grad_scale), int(max_blocks),
              _p(gate), _p(hyper), _stream())


def sumsq_workspace_floats() -> int:
    return int(_lib.load().afk_sumsq_workspace_floats())


def sumsq_(x, acc, *, gate=None, ws=None):
    """acc[0] += sum(x^2) over a bf16 range (fp32, deterministic); skipped on the device when gate (int32[1]) reads 0"""
    _chk(x, BF16, "sumsq x")
    if ws is None:
        ws = torch.empty(_lib.load().afk_sumsq_workspace_floats(), device=x.device, dtype=torch.float32)
    _lib.call("afk_sumsq_bf16", x.data_ptr(), x.numel(), _chk(acc, torch.float32).data_ptr(), _p(gate), ws.data_ptr(), _stream())
    return acc


def clip_coef_(sumsq, coef, *, max_norm, scale=1.0, norm_out=None):
    """coef[0] = min(1, max_norm / (sqrt(sum(sumsq)) * scale + 1e-6))   (torch.nn.utils.clip_grad_norm_); norm_out[0] = the norm.
    sumsq: float32 [n] partial sums, folded in index order"""
    _lib.call("afk_clip_coef", _chk(sumsq, torch.float32).data_ptr(), sumsq.numel(), float(scale), float(max_norm), coef.data_ptr(), _p(norm_out), _stream())


def set_f32(dst, values):
    """dst[:len(values)] = values (<= 4 floats), by a kernel launch on the current stream"""
    v = [float(x) for x in values] + [0.0] * (4 - len(values))
    _lib.call("afk_set_f32", _chk(dst, torch.float32).data_ptr(), len(values), v[0], v[1], v[2], v[3], _stream())


def gemm_set_variant(v: int):
    """0 auto, 1 = 128x128 kernel, 2 = 256x256 ping-pong kernel (tests / microbenchmarks)"""
    _lib.call("afk_gemm_set_variant", int(v))


# ---------------------------------------------------------------------------------------------- profiling
KERNEL_FAMILIES = ("gemm_nt128", "gemm_nt256", "gemm_nn256", "gemm_tn256", "gemm_splitk", "gemv", "attn2_fwd_d64", "attn2_fwd_d128",
                   "attn2_bwd_d64", "attn2_bwd_d128", "gqa_reduce", "xattn_fwd", "xattn_bwd", "attn1_fwd", "attn1_bwd", "gemm_generic_epilogue")


def kernel_counts(reset: bool = False) -> dict:
    """launches per kernel family since the last reset (include/afk.h AFK_CNT_*): lets a test assert which kernel served a shape"""
    import ctypes

    buf = (ctypes.c_int64 * len(KERNEL_FAMILIES))()
    _lib.call("afk_kernel_counts", ctypes.addressof(buf), len(KERNEL_FAMILIES))
    if reset:
        _lib.call("afk_kernel_counts_reset")
    return dict(zip(KERNEL_FAMILIES, list(buf)))


def prof_enable(on: bool):
    _lib.call("afk_prof_enable", int(on))


def prof_reset():
    _lib.call("afk_prof_reset")


def prof_collect():
    import ctypes
    ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    _lib.call("afk_prof_collect", ctypes.addressof(ms), ctypes.addressof(fl), ctypes.addressof(n))
    return ms.value, fl.value, n.value
