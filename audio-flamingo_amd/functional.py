"""Layer-level autograd Functions with hand-written backward passes.

Each Function is one stage of the oracle's forward expressed as a sequence of afk kernels, and
its backward is the matching hand-derived sequence: autograd only stitches the stages together.  Weight
gradients never travel through autograd: the wgrad GEMMs write (or accumulate) directly into the gradient arena
(arena.py) and the Functions return ``None`` for their parameter inputs.  The single ``anchor`` tensor each
Function takes is a parameter view that only serves to make the output require grad.

Backward GEMM forms (default "wgrad_direct", see BWD_FORM below):
    dgrad  dX[M,K] = dY[M,N] . Wt[K,N]^T          NT kernel; Wt = arena shadow (W^T, refreshed after every optimizer step)
    wgrad  dW[N,K] = dY[M,N]^T . X[M,K]           TN kernel (csrc/gemm256t.hip) on the operands as they lie; narrow outputs (< 192 tiles
                                                  of 256x256) fall back to NT on afk_transpose_bf16 copies (dYt . Xt^T)
    bias   db[N]   = colsum(dY)  (rowsum(dYt) on the fallback)
"""
from __future__ import annotations

import os

import torch

from . import ops
from .arena import Arena, Block


# ---------------------------------------------------------------------------------------------- helpers
def _wgrad(arena: Arena, blk: Block, dyt, xt, Mvalid, *, bias_blk=None, bias_slices=None, dyt_rows=None):
    """blk.grad[N,K] (+)= dyt[N,Mp] . xt[K,Mp]^T ; bias grad(s) from rowsum(dyt)."""
    gw = blk.grad.reshape(blk.shape[0], -1)
    ops.gemm_nt(dyt, xt, out=gw, accumulate=not blk.fresh)
    arena.grad_written(blk)
    if bias_blk is not None:
        acc = not bias_blk.fresh
        if bias_slices is None:
            ops.rowsum(dyt, Mvalid, bias_blk.grad, accumulate=acc)
        else:
            for (s, e) in bias_slices:
                ops.rowsum(dyt[s:e], Mvalid, bias_blk.grad[s:e], accumulate=acc)
        arena.grad_written(bias_blk)


# Backward GEMM forms (measured on the full AF3-7B step, same box, ms/step):
#   "nt"            every backward GEMM on the NT kernel: dY^T / X^T operand transposes for wgrad + W^T shadows for dgrad        455-459
#   "wgrad_direct"  wgrad on the TN kernel straight from dY and X (no activation transposes), dgrad on NT + shadows   (default)  442
#   "direct"        also dgrad on the NN kernel (no W^T shadows: -15.6 GB of HBM), the NN kernel is ~6 % slower than NT           460
# The transposed-operand kernels have no small-tile variant: outputs with fewer than DIRECT_MIN_TILES 256x256 tiles stay on "nt".
BWD_FORM = os.environ.get("AFK_BWD_FORM", "wgrad_direct")
DIRECT_MIN_TILES = 192


def _tiles256(m, n):
    return ((m + 255) // 256) * ((n + 255) // 256)


# SwiGLU backward inside the down-projection dgrad epilogue (AFK_GEMM_SWIGLU_BWD): bit-identical and saves 620 MB of HBM traffic per
# layer, but MEASURED SLOWER on the full step (446-450 vs 435 ms): the epilogue's per-lane row-strided 8-byte gate|up loads cost more
# than the separate 5.3 TB/s elementwise pass.  Kept as an ABI feature (tests/test_ops_gpu.py), off in the model.
FUSE_SWIGLU_BWD = os.environ.get("AFK_FUSE_SWIGLU", "0") == "1"
# SwiGLU forward inside the gate|up GEMM (AFK_GEMM_SWIGLU_FWD): a workgroup computes 128 gate and the matching 128 up features, the up
# waves hand their bf16 tile to the gate waves through the idle operand buffers, and silu(g) * u is stored beside g|u - bit-identical to
# the separate silu_mul_fwd pass, whose 0.93 GB of HBM traffic per layer disappears.
FUSE_SWIGLU_FWD = os.environ.get("AFK_FUSE_SWIGLU_FWD", "1") == "1"
# GELU outputs (encoder fc1, projector linear_1, conv1) are KEPT for backward instead of being recomputed from the saved pre-activation by a
# separate gelu_fwd pass (round 2): 123 MB per encoder layer at B = 8 (3.9 GB for the tower, of 288 GB) buys back 34 launches and
# 1.6 ms per step; bit-identical (the same kernel produced the same values either way).  AFK_SAVE_GELU=0 restores the recompute.
SAVE_GELU = os.environ.get("AFK_SAVE_GELU", "1") == "1"
LMHEAD_NN_DGRAD = os.environ.get("AFK_LMHEAD_NN_DGRAD", "1") == "1"   # see LMHeadLossFn.forward
# Per-weight choice of the dgrad form: weights whose key ends with one of these suffixes take dX = dY . W on the transposed-operand (NN) kernel
# straight from W - their W^T shadow turns lazy and is no longer refreshed every step - the others keep the NT kernel + shadow ("direct" = every
# weight).  Measured on the full step, alternating runs in one gpurun call (profiles/r03_ab_experiments.md §4): gate|up alone 409.1 vs 414.6 ms
# (three pairs: -5.0 / -4.6 / -6.9 ms; its shadow is the largest - 271 MB per layer, 3 ms of transposes per step - and its dgrad reduces over
# K = 37 888, where the NN kernel is level with NT); adding down_proj (+0.4 / +4 ms), o_proj + qkv (+0.1) or the encoder weights (-0.7) on top of it
# is neutral or worse.  Round 1-2 measured "direct" (all weights) slower and stopped there.
NN_DGRAD_DEFAULT = "mlp.gate_up.weight"
NN_DGRAD_SUFFIXES = tuple(x for x in os.environ.get("AFK_NN_DGRAD", NN_DGRAD_DEFAULT).split(",") if x)


# Recomputation of a checkpointed layer (torch.utils.checkpoint re-runs the stage's forward inside backward; modeling._layer sets the flag around exactly that
# re-run): what the re-run produces is only read through the tensors the stage KEEPS for its backward - its output is thrown away - so the GEMM that produces
# nothing but the output (encoder fc2, decoder down_proj: 36 % / 29 % of a layer's forward GEMM work) is dead there and is skipped.  The reference's
# GradientCheckpointingLayer recomputes it (autograd cannot know); gradients are bit-identical either way (AFK_RECOMPUTE_SKIP_TAIL=0 restores it).
RECOMPUTING = False
RECOMPUTE_SKIP_TAIL = os.environ.get("AFK_RECOMPUTE_SKIP_TAIL", "1") == "1"


class recomputing:
    """context manager of the re-run (the second manager of torch.utils.checkpoint's context_fn)"""

    def __enter__(self):
        global RECOMPUTING
        self.prev, RECOMPUTING = RECOMPUTING, True

    def __exit__(self, *a):
        global RECOMPUTING
        RECOMPUTING = self.prev


# Bias gradients where their operand is produced (round 6, AFK_FUSE_BIAS_SUMS=0 restores the separate column-sum passes): the column-owned LayerNorm backward
# also sums the residual-stream gradient it writes (= grad_output of the Linear below the norm: out_proj in the same layer, fc2 of the LOWER layer), and the GELU
# backward sums the d(pre-activation) it writes (= grad_output of fc1).
FUSE_BIAS_SUMS = os.environ.get("AFK_FUSE_BIAS_SUMS", "1") == "1"


def linear_bwd(arena: Arena, dy, x, wkey, *, bkey=None, bias_slices=None, need_dx=True, swiglu_gu=None, bias_presum=None):
    """y = x W^T (+ b):  returns dx, writes dW (and db) into the arena.
    The weight-gradient branch (two operand transposes + wgrad GEMM + bias row-sum) is independent of the data-gradient
    GEMM; with ``arena.wgrad_stream`` set it is enqueued on that stream so the two GEMMs fill each other's tile tails."""
    # bias_presum: dy's producer has already summed its columns - "done": into the bias gradient itself (accumulate semantics applied there);
    # a bf16 [N] tensor: into a scratch row that the weight-gradient branch copies / adds into the bias gradient (no column-sum pass either way)
    M = dy.shape[0]
    blk = arena[wkey]
    side = arena.wgrad_stream
    N, K = dy.shape[1], x.shape[1]
    presum = None
    if bias_presum is not None:
        assert bkey is not None and bias_slices is None
        if isinstance(bias_presum, str):
            arena.grad_written(arena[bkey])
        else:
            presum = (arena[bkey], bias_presum)
        bkey = None

    def take_presum():
        if presum is not None:
            bb0, row = presum
            if bb0.fresh:
                bb0.grad.copy_(row)
            else:
                bb0.grad.add_(row)
            arena.grad_written(bb0)
    direct = BWD_FORM == "direct"
    wdirect = direct or BWD_FORM == "wgrad_direct"  # TN wgrad only: no activation transposes, dgrad stays on the NT kernel + W^T shadows

    def wgrad_branch():
        take_presum()
        if wdirect and _tiles256(N, K) * ops.splitk_plan_256(N, K, M) >= DIRECT_MIN_TILES:  # narrow outputs: TN kernel with split-K
            ops.gemm(dy, x, out=blk.grad.reshape(blk.shape[0], -1), trans_a=True, trans_b=True, accumulate=not blk.fresh)
            arena.grad_written(blk)
            if bkey:
                bb = arena[bkey]
                if bias_slices is not None and len(bias_slices) == 2 and bias_slices[0][0] == 0 and bias_slices[1][1] == N:
                    # two slices with a gap (the encoder's fused q|k|v bias: k_proj has no bias, :112): ONE column-sum launch over the full width,
                    # then the gap is cleared - its gradient must read exactly zero so that the k-third of the fused bias never moves
                    ops.colsum(dy, bb.grad, accumulate=not bb.fresh)
                    bb.grad[bias_slices[0][1]: bias_slices[1][0]].zero_()
                else:
                    for (s, e) in (bias_slices or [(0, N)]):
                        ops.colsum(dy[:, s:e], bb.grad[s:e], accumulate=not bb.fresh)
                arena.grad_written(bb)
            return
        dyt = ops.transpose(dy)  # [N, Mp]
        xt = ops.transpose(x)    # [K, Mp]
        _wgrad(arena, blk, dyt, xt, M, bias_blk=arena[bkey] if bkey else None, bias_slices=bias_slices)

    if side is None:
        wgrad_branch()
    else:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            side.wait_event(ev)
            ops.THIN_BLOCKS = arena.thin_blocks  # operand transposes trickle beside the dgrad GEMM instead of evicting it
            try:
                wgrad_branch()
            finally:
                ops.THIN_BLOCKS = 0
        dy.record_stream(side)
        x.record_stream(side)
        if presum is not None:
            presum[1].record_stream(side)
    if not need_dx:
        return None
    if (direct or (NN_DGRAD_SUFFIXES and wkey.endswith(NN_DGRAD_SUFFIXES) and swiglu_gu is None)) and _tiles256(M, K) >= DIRECT_MIN_TILES:
        if not direct:
            blk.shadow_lazy = True   # nobody reads this W^T shadow in the step any more: skip its per-step refresh
        return ops.gemm(dy, blk.data.reshape(blk.shape[0], -1), trans_b=True)  # dX = dY . W, W as stored
    wt = arena.shadow(wkey)  # [K, pad64(N)]
    if swiglu_gu is not None:  # down-projection dgrad with the SwiGLU backward in its epilogue: returns d(gate|up) directly
        return ops.gemm_nt(dy, wt, K=dy.shape[1], swiglu_bwd=swiglu_gu)
    return ops.gemm_nt(dy, wt, K=dy.shape[1])


# ---------------------------------------------------------------------------------------------- conv stem (a2)
class ConvStemFn:
    """gelu(conv1) -> gelu(conv2, stride 2) -> permute -> + embed_positions   (modeling_audioflamingo3.py:380-385)"""

    @staticmethod
    def forward(ctx, feats, anchor, arena, keys, pos, W, T, C):
        k1w, k1b, k2w, k2b = keys
        E = arena[k1w].shape[0]
        col1 = ops.im2col_conv1(feats)  # [W*T, 3C]
        w1 = arena.shadow(k1w)  # [E, 3C] tap-major
        M1 = W * T
        pre1 = torch.empty((M1, E), device=feats.device, dtype=torch.bfloat16)
        h1 = ops.gemm_nt(col1, w1, bias=arena[k1b].data, gelu=True, preact_out=pre1)
        T2 = (T - 1) // 2 + 1
        col2 = ops.im2col_conv2(h1, W, T, E)
        del h1
        w2 = arena.shadow(k2w)
        pre2 = torch.empty((W * T2, E), device=feats.device, dtype=torch.bfloat16)
        x0 = ops.gemm_nt(col2, w2, bias=arena[k2b].data, gelu=True, preact_out=pre2, residual=pos, res_mod=T2)
        ctx.save_for_backward(col1, pre1, pre2, col2 if SAVE_GELU else None)   # col2 = im2col(gelu(pre1)): kept, or rebuilt in backward
        ctx.meta = (arena, keys, W, T, T2, E)
        return x0

    @staticmethod
    def backward(ctx, dx0):
        col1, pre1, pre2, col2 = ctx.saved_tensors
        arena, (k1w, k1b, k2w, k2b), W, T, T2, E = ctx.meta
        dx0 = dx0.contiguous()
        dpre2 = ops.gelu_bwd(dx0, pre2)
        if col2 is None:
            h1 = ops.gelu_fwd(pre1)
            col2 = ops.im2col_conv2(h1, W, T, E)
            del h1
        dyt = ops.transpose(dpre2)
        xt = ops.transpose(col2)
        del col2
        b2 = arena[k2w]
        dwp = ops.gemm_nt(dyt, xt)  # [E, 3E] tap-major
        ops.conv_weight_grad_from_gemm(dwp, b2.grad, accumulate=not b2.fresh)
        arena.grad_written(b2)
        bb = arena[k2b]
        ops.rowsum(dyt, W * T2, bb.grad, accumulate=not bb.fresh)
        arena.grad_written(bb)
        del dyt, xt, dwp
        dcol2 = ops.gemm_nt(dpre2, arena.shadow_aux(k2w), K=E)  # [W*T2, 3E]
        dh1 = ops.col2im_conv2(dcol2, W, T, E)
        del dcol2
        dpre1 = ops.gelu_bwd(dh1, pre1)
        del dh1
        dyt = ops.transpose(dpre1)
        xt = ops.transpose(col1)
        b1 = arena[k1w]
        dwp = ops.gemm_nt(dyt, xt)
        ops.conv_weight_grad_from_gemm(dwp, b1.grad, accumulate=not b1.fresh)
        arena.grad_written(b1)
        bb = arena[k1b]
        ops.rowsum(dyt, W * T, bb.grad, accumulate=not bb.fresh)
        arena.grad_written(bb)
        return None, None, None, None, None, None, None, None


# ---------------------------------------------------------------------------------------------- encoder layer (a5, a6)
def _lower_fc2_bias(arena, pfx):
    """pfx = '...layers.<i>.': the key of layer i - 1's fc2 bias (the consumer of the column sums of this layer's dx), or None"""
    head, _, tail = pfx.rstrip(".").rpartition(".")
    key = f"{head}.{int(tail) - 1}.fc2.bias" if tail.isdigit() and int(tail) > 0 else None
    return key if key in arena.blocks else None


class EncoderLayerFn:
    """pre-LN attention block + pre-LN GELU MLP   (AudioFlamingo3EncoderLayer.forward, :211-245)"""

    @staticmethod
    def forward(ctx, x, anchor, arena, pfx, W, S, H, kv_len):
        E = x.shape[1]
        D = E // H
        A = lambda k: arena[pfx + k]
        h, mean1, rstd1 = ops.layernorm_fwd(x, A("self_attn_layer_norm.weight").data, A("self_attn_layer_norm.bias").data)
        qkv = ops.gemm_nt(h, A("self_attn.qkv.weight").data, bias=A("self_attn.qkv.bias").data)
        # q*scaling before the dot product (:142) == scaling the scores by the same power of two
        o, lse = ops.attn_fwd(qkv, W, S, H, H, D, scale=D ** -0.5, causal=False, kv_len=kv_len)
        x2 = ops.gemm_nt(o, A("self_attn.out_proj.weight").data, bias=A("self_attn.out_proj.bias").data, residual=x)
        h2, mean2, rstd2 = ops.layernorm_fwd(x2, A("final_layer_norm.weight").data, A("final_layer_norm.bias").data)
        pre = torch.empty((x.shape[0], A("fc1.weight").shape[0]), device=x.device, dtype=torch.bfloat16)
        f = ops.gemm_nt(h2, A("fc1.weight").data, bias=A("fc1.bias").data, gelu=True, preact_out=pre)
        if RECOMPUTING and RECOMPUTE_SKIP_TAIL:
            x3 = torch.empty_like(x2)   # the re-run's output is discarded: fc2 is dead work there
        else:
            x3 = ops.gemm_nt(f, A("fc2.weight").data, bias=A("fc2.bias").data, residual=x2)
        ctx.save_for_backward(x, mean1, rstd1, h, qkv, o, lse, x2, mean2, rstd2, h2, pre, kv_len, f if SAVE_GELU else None)
        ctx.meta = (arena, pfx, W, S, H, D)
        return x3

    @staticmethod
    def backward(ctx, dx3):
        x, mean1, rstd1, h, qkv, o, lse, x2, mean2, rstd2, h2, pre, kv_len, f = ctx.saved_tensors
        arena, pfx, W, S, H, D = ctx.meta
        E = x.shape[1]
        A = lambda k: arena[pfx + k]
        dx3 = dx3.contiguous()
        if f is None:
            f = ops.gelu_fwd(pre)
        fuse = FUSE_BIAS_SUMS and E % 8 == 0 and E <= 4096
        # fc2's grad_output is the gradient this layer received: if the layer above produced it (same storage), its LayerNorm backward has summed its columns already
        pre2 = None
        if fuse and arena.presums.get("for") == pfx + "fc2.bias" and arena.presums.get("ptr") == dx3.data_ptr():
            pre2 = arena.presums["row"]
        arena.presums.clear()   # one slot, consumed or dropped by the next layer down: nothing stale survives to a later backward
        df = linear_bwd(arena, dx3, f, pfx + "fc2.weight", bkey=pfx + "fc2.bias", bias_presum=pre2)
        del f
        b1 = A("fc1.bias")
        if fuse and pre.shape[1] % 8 == 0:
            dpre = ops.gelu_bwd(df, pre, colsum_out=b1.grad, colsum_accumulate=not b1.fresh)
            done1 = "done"
        else:
            dpre, done1 = ops.gelu_bwd(df, pre), None
        del df
        dh2 = linear_bwd(arena, dpre, h2, pfx + "fc1.weight", bkey=pfx + "fc1.bias", bias_presum=done1)
        del dpre
        lw, lb = A("final_layer_norm.weight"), A("final_layer_norm.bias")
        bo = A("self_attn.out_proj.bias")
        dx2 = ops.layernorm_bwd(x2, lw.data, dh2, mean2, rstd2, lw.grad, lb.grad, dx_add=dx3, accumulate=not lw.fresh,
                                colsum_out=bo.grad if fuse else None, colsum_accumulate=not bo.fresh)
        arena.grad_written(lw), arena.grad_written(lb)
        del dh2
        do = linear_bwd(arena, dx2, o, pfx + "self_attn.out_proj.weight", bkey=pfx + "self_attn.out_proj.bias", bias_presum="done" if fuse else None)
        dqkv = ops.attn_bwd(qkv, o, do, lse, W, S, H, H, D, scale=D ** -0.5, causal=False, kv_len=kv_len)
        del do
        # k_proj has no bias (:112): only the q and v thirds of the fused bias receive a gradient
        dh = linear_bwd(arena, dqkv, h, pfx + "self_attn.qkv.weight", bkey=pfx + "self_attn.qkv.bias",
                        bias_slices=[(0, E), (2 * E, 3 * E)])
        del dqkv
        lw, lb = A("self_attn_layer_norm.weight"), A("self_attn_layer_norm.bias")
        # the dx this layer hands down is the grad_output of the LOWER layer's fc2: its column sums go into a scratch row, registered under dx's storage - the
        # lower layer takes them only if it receives exactly this tensor (anything else - another consumer of the hidden state, a copy - falls back to the column-sum pass)
        lower = _lower_fc2_bias(arena, pfx) if fuse else None
        row = torch.empty(E, device=x.device, dtype=torch.bfloat16) if lower else None
        dx = ops.layernorm_bwd(x, lw.data, dh, mean1, rstd1, lw.grad, lb.grad, dx_add=dx2, accumulate=not lw.fresh, colsum_out=row, colsum_accumulate=False)
        arena.grad_written(lw), arena.grad_written(lb)
        if row is not None:
            arena.presums.update({"for": lower, "ptr": dx.data_ptr(), "row": row})
        return dx, None, None, None, None, None, None, None


# ---------------------------------------------------------------------------------------------- avg-pool + LN (a7)
class PoolNormFn:
    """AvgPool1d(2,2) over time then LayerNorm (:401-403)"""

    @staticmethod
    def forward(ctx, x, anchor, arena, wkey, bkey, out_rows):
        E = x.shape[1]
        pooled = ops.avgpool2_fwd(x, out_rows, E)
        y, mean, rstd = ops.layernorm_fwd(pooled, arena[wkey].data, arena[bkey].data)
        ctx.save_for_backward(pooled, mean, rstd)
        ctx.meta = (arena, wkey, bkey, out_rows, E)
        return y

    @staticmethod
    def backward(ctx, dy):
        pooled, mean, rstd = ctx.saved_tensors
        arena, wkey, bkey, out_rows, E = ctx.meta
        lw, lb = arena[wkey], arena[bkey]
        dp = ops.layernorm_bwd(pooled, lw.data, dy.contiguous(), mean, rstd, lw.grad, lb.grad, accumulate=not lw.fresh)
        arena.grad_written(lw), arena.grad_written(lb)
        return ops.avgpool2_bwd(dp, out_rows, E), None, None, None, None, None


class RotaryTimeFn:
    """Music Flamingo: rotary time embedding on the encoder output rows (apply_rotary_time_emb, modeling_musicflamingo.py:187-204)"""

    @staticmethod
    def forward(ctx, x, cos, sin, arena):
        ctx.save_for_backward(cos, sin)
        return ops.rotary_time(x, cos, sin)

    @staticmethod
    def backward(ctx, dy):
        cos, sin = ctx.saved_tensors
        return ops.rotary_time(dy.contiguous(), cos, sin, backward=True), None, None, None


# ---------------------------------------------------------------------------------------------- projector (a8)
class ProjectorFn:
    """Linear -> GELU -> Linear (AudioFlamingo3MultiModalProjector.forward, :435-439)"""

    @staticmethod
    def forward(ctx, x, anchor, arena, pfx):
        A = lambda k: arena[pfx + k]
        pre = torch.empty((x.shape[0], A("linear_1.weight").shape[0]), device=x.device, dtype=torch.bfloat16)
        a = ops.gemm_nt(x, A("linear_1.weight").data, bias=A("linear_1.bias").data, gelu=True, preact_out=pre)
        y = ops.gemm_nt(a, A("linear_2.weight").data, bias=A("linear_2.bias").data)
        ctx.save_for_backward(x, pre, a if SAVE_GELU else None)
        ctx.meta = (arena, pfx)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, pre, a = ctx.saved_tensors
        arena, pfx = ctx.meta
        dy = dy.contiguous()
        if a is None:
            a = ops.gelu_fwd(pre)
        da = linear_bwd(arena, dy, a, pfx + "linear_2.weight", bkey=pfx + "linear_2.bias")
        dpre = ops.gelu_bwd(da, pre)
        dx = linear_bwd(arena, dpre, x, pfx + "linear_1.weight", bkey=pfx + "linear_1.bias")
        return dx, None, None, None


# ---------------------------------------------------------------------------------------------- embedding scatter (a10)
class EmbedScatterFn:
    """embed_tokens(input_ids) with <sound> rows overwritten by audio rows in row-major order (:532-545)"""

    @staticmethod
    def forward(ctx, audio, anchor, arena, ekey, ids, src):
        out = ops.embed_scatter_fwd(ids, src, arena[ekey].data, audio)
        ctx.save_for_backward(ids, src)
        ctx.meta = (arena, ekey, None if audio is None else audio.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        ids, src = ctx.saved_tensors
        arena, ekey, ashape = ctx.meta
        blk = arena[ekey]
        if blk.fresh:
            blk.grad.zero_()  # sparse row updates need a cleared destination
        d_audio = None
        if ashape is not None:
            # rows of padded windows that no placeholder references keep a zero gradient
            d_audio = torch.zeros(ashape, device=dout.device, dtype=torch.bfloat16)
        ops.embed_scatter_bwd(ids, src, dout.contiguous(), blk.grad, d_audio)
        arena.grad_written(blk)
        return d_audio, None, None, None, None, None


# ---------------------------------------------------------------------------------------------- decoder layer (a13-a16)
class DecoderLayerFn:
    """RMSNorm -> GQA causal attention (RoPE) -> +res ; RMSNorm -> SwiGLU -> +res  (Qwen2DecoderLayer.forward, :269-298)"""

    @staticmethod
    def forward(ctx, x, anchor, arena, pfx, B, S, Hq, Hkv, D, eps, cos, sin, pos, kv_len, krange=None, kv_lo=None, rows=None):
        # rows (int64, ascending, on the device; None = every row; round 6): the LAST decoder layer of a training step with labels needs its residual stream only
        # at the positions the loss reads (their keys / values still come from every row): behind the attention, o_proj, the post-attention norm and the MLP run
        # on the gathered rows alone and the layer returns [len(rows), H].  Same loss and - up to the summation order inside the GEMMs - the same gradients (a
        # row the loss does not read has an exactly zero grad_output); the reference computes every row (modeling_qwen2.py:269-298) and drops them at the loss.
        A = lambda k: arena[pfx + k]
        h, rstd1 = ops.rmsnorm_fwd(x, A("input_layernorm.weight").data, eps)
        # qkv projection with the rotary embedding in the GEMM's epilogue where the shape allows (ops.gemm_nt_rope, round 6)
        qkv = ops.gemm_nt_rope(h, A("self_attn.qkv.weight").data, A("self_attn.qkv.bias").data, cos, sin, S=S, rope_cols=(Hq + Hkv) * D, D=D, pos=pos)
        if krange is not None:  # left-padded rows (the reference processor's default): per-query key intervals [lo_b, min(i + 1, hi_b))
            o, lse = ops.attn_interval_fwd(qkv, krange, B, S, Hq, Hkv, D, scale=D ** -0.5)
        else:
            # right padding: kv_len; left padding on the LDS kernels (head_dim 64 / 128): kv_lo beside it
            o, lse = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=True, kv_len=kv_len, kv_lo=kv_lo)
        if rows is not None:
            x2 = ops.gemm_nt(ops.gather_rows(o, rows), A("self_attn.o_proj.weight").data, residual=ops.gather_rows(x, rows))
        else:
            x2 = ops.gemm_nt(o, A("self_attn.o_proj.weight").data, residual=x)
        h2, rstd2 = ops.rmsnorm_fwd(x2, A("post_attention_layernorm.weight").data, eps)
        wgu = A("mlp.gate_up.weight").data
        if FUSE_SWIGLU_FWD and wgu.shape[0] % 256 == 0:
            a = torch.empty((h2.shape[0], wgu.shape[0] // 2), device=h2.device, dtype=torch.bfloat16)
            gu = ops.gemm_nt(h2, wgu, swiglu_fwd_out=a)   # gate|up and silu(gate) * up from one launch
        else:
            gu = ops.gemm_nt(h2, wgu)
            a = ops.silu_mul_fwd(gu)
        if RECOMPUTING and RECOMPUTE_SKIP_TAIL:
            x3 = torch.empty_like(x2)   # the re-run's output is discarded: down_proj is dead work there
        else:
            x3 = ops.gemm_nt(a, A("mlp.down_proj.weight").data, residual=x2)
        # `a` (310 MB / layer at B=8) is kept: 288 GB of HBM makes the recompute pass the worse trade
        ctx.save_for_backward(x, rstd1, h, qkv, o, lse, x2, rstd2, h2, gu, cos, sin, pos, kv_len, a, krange, kv_lo, rows)
        ctx.meta = (arena, pfx, B, S, Hq, Hkv, D)
        return x3

    @staticmethod
    def backward(ctx, dx3):
        x, rstd1, h, qkv, o, lse, x2, rstd2, h2, gu, cos, sin, pos, kv_len, a, krange, kv_lo, rows = ctx.saved_tensors
        arena, pfx, B, S, Hq, Hkv, D = ctx.meta
        A = lambda k: arena[pfx + k]
        dx3 = dx3.contiguous()
        if FUSE_SWIGLU_BWD and BWD_FORM != "direct":
            dgu = linear_bwd(arena, dx3, a, pfx + "mlp.down_proj.weight", swiglu_gu=gu)
            del a
        else:
            da = linear_bwd(arena, dx3, a, pfx + "mlp.down_proj.weight")
            del a
            dgu = ops.silu_mul_bwd(gu, da)
            del da
        dh2 = linear_bwd(arena, dgu, h2, pfx + "mlp.gate_up.weight")
        del dgu
        nw = A("post_attention_layernorm.weight")
        dx2 = ops.rmsnorm_bwd(x2, nw.data, dh2, rstd2, nw.grad, dx_add=dx3, accumulate=not nw.fresh)
        arena.grad_written(nw)
        del dh2
        if rows is not None:   # the rows the loss reads: their gradients go back to their places, every other row of d(attention output) / d(residual) is zero
            do = ops.scatter_rows(linear_bwd(arena, dx2, ops.gather_rows(o, rows), pfx + "self_attn.o_proj.weight"), rows, x.shape[0])
            dx2 = ops.scatter_rows(dx2, rows, x.shape[0])
        else:
            do = linear_bwd(arena, dx2, o, pfx + "self_attn.o_proj.weight")
        if krange is not None:
            dqkv = ops.attn_interval_bwd(qkv, o, do, lse, krange, B, S, Hq, Hkv, D, scale=D ** -0.5)
        else:
            # the rotary backward rides in the dQ epilogue / the GQA reduce (ops.attn_bwd(rope=...), round 6): no separate pass over dq | dk
            dqkv = ops.attn_bwd(qkv, o, do, lse, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=True, kv_len=kv_len, kv_lo=kv_lo, rope=(cos, sin, pos))
        del do
        if krange is not None:
            ops.rope_(dqkv, cos, sin, S=S, nheads=Hq + Hkv, D=D, pos=pos, backward=True)
        dh = linear_bwd(arena, dqkv, h, pfx + "self_attn.qkv.weight", bkey=pfx + "self_attn.qkv.bias")
        del dqkv
        nw = A("input_layernorm.weight")
        dx = ops.rmsnorm_bwd(x, nw.data, dh, rstd1, nw.grad, dx_add=dx2, accumulate=not nw.fresh)
        arena.grad_written(nw)
        return (dx,) + (None,) * 16


class RMSNormFn:
    """final Qwen2RMSNorm (modeling_qwen2.py:398)"""

    @staticmethod
    def forward(ctx, x, anchor, arena, wkey, eps):
        y, rstd = ops.rmsnorm_fwd(x, arena[wkey].data, eps)
        ctx.save_for_backward(x, rstd)
        ctx.meta = (arena, wkey)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, rstd = ctx.saved_tensors
        arena, wkey = ctx.meta
        nw = arena[wkey]
        dx = ops.rmsnorm_bwd(x, nw.data, dy.contiguous(), rstd, nw.grad, accumulate=not nw.fresh)
        arena.grad_written(nw)
        return dx, None, None, None, None


# ---------------------------------------------------------------------------------------------- lm_head (+ fused loss) (a17, a18)
class LMHeadFn:
    """logits = hidden @ lm_head.weight^T   (materialised; used for generate() and parity checks)"""

    @staticmethod
    def forward(ctx, x, anchor, arena, wkey):
        y = ops.gemm_nt(x, arena[wkey].data)
        ctx.save_for_backward(x)
        ctx.meta = (arena, wkey)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        arena, wkey = ctx.meta
        return linear_bwd(arena, dy.contiguous(), x, wkey), None, None, None


class LMHeadLossFn:
    """lm_head + shifted cross-entropy without materialising [B*S, V] (lm_head :625-627 + ForCausalLMLoss, loss_utils.py:51-72).

    Row chunks: logits chunk (bf16, as the oracle's lm_head output) -> afk_ce_fwd_bwd turns it into dlogits in place
    -> dgrad / wgrad GEMMs consume it -> the chunk buffer is reused.  Gradients are produced here for an upstream
    gradient of 1 and rescaled by the device-side grad_output in backward (no host sync).
    """

    CHUNK = 4096  # 16x14 = 224 tiles of 256x256 for the dgrad GEMM: enough to fill the chip with the fast kernel

    @staticmethod
    def key_state(arena, wkey):
        """part of the stage key (stage_ops.py): whether the unscaled dW is parked in the gradient arena itself (fresh block) or in a private buffer
        decides what forward keeps for backward"""
        return f"fresh{int(arena[wkey].fresh)}"

    @staticmethod
    def forward(ctx, x, anchor, arena, wkey, shift_labels, denom, rows=None):
        # rows (int64, ascending, on the device; None = every row): the positions whose shifted label is not -100.  Only they
        # contribute to the loss and to any gradient (the CE gradient of an ignored row is exactly zero), so the lm_head GEMM, the CE
        # kernel and both backward GEMMs run on the gathered rows alone: identical loss and gradients, 1/4 of the work on the
        # 256-answer-token batches of the benchmark.  The reference computes all B*S rows (modeling_audioflamingo3.py:625-633).
        M_all = x.shape[0]
        if rows is not None:
            x = ops.gather_rows(x, rows)
            shift_labels = shift_labels.index_select(0, rows)
        M, H = x.shape
        blk = arena[wkey]
        V = blk.shape[0]
        dev = x.device
        need_grad = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1])  # grad mode is off inside forward()
        row_loss = torch.empty(M, device=dev, dtype=torch.float32)
        dx = torch.empty_like(x) if need_grad else None
        chunk = min(LMHeadLossFn.CHUNK, M)
        buf = torch.empty((chunk, V), device=dev, dtype=torch.bfloat16)
        direct = BWD_FORM == "direct"
        wdirect = direct or BWD_FORM == "wgrad_direct"
        # dgrad dX[n, H] = dlogits[n, V] . W[V, H] reduces over the VOCABULARY (152 064) into a narrow output (n x 3584: 112 tiles of 256 x 256 for
        # the 2 048 labelled rows of the benchmark batch): the NT form + W^T shadow ran it on the 128-tile kernel at 0.88 PF/s.  LMHEAD_NN_DGRAD
        # runs it on the transposed-operand kernel straight from W with split-K over the vocabulary (one round of 224 workgroups, fp32 partials
        # folded in fixed order): no W^T shadow of lm_head is read by the step any more, so its 1.09 GB refresh per step is skipped too
        # (the block's shadow turns lazy; LMHeadFn's backward still rebuilds it on demand).
        nn_dgrad = LMHEAD_NN_DGRAD and not direct and H % 8 == 0 and _tiles256(chunk, H) < DIRECT_MIN_TILES
        if nn_dgrad and not blk.shadow_lazy:
            blk.shadow_lazy = True
        wt = arena.shadow(wkey) if need_grad and not direct and not nn_dgrad else None
        if need_grad:
            # unscaled lm_head gradient goes to a private buffer when it must be accumulated into existing grads
            gw_tmp = blk.grad if blk.fresh else torch.empty_like(blk.grad)
        first = True
        for s in range(0, M, chunk):
            e = min(M, s + chunk)
            n = e - s
            logits = buf[:n]
            ops.gemm_nt(x[s:e], blk.data, out=logits)
            ops.ce_fwd_bwd_(logits, shift_labels[s:e], row_loss[s:e], denom, upstream=1.0, write_grad=need_grad)
            if need_grad:
                if direct or nn_dgrad:
                    ops.gemm(logits, blk.data, out=dx[s:e], trans_b=True)                               # dX = dlogits . W  (split-K chosen by ops.splitk_plan_256)
                else:
                    ops.gemm_nt(logits, wt, out=dx[s:e], K=V)
                if wdirect:
                    ops.gemm(logits, x[s:e], out=gw_tmp, trans_a=True, trans_b=True, accumulate=not first)  # dW += dlogits^T . X
                else:
                    dlt = ops.transpose(logits)      # [V, pad64(n)]
                    xt = ops.transpose(x[s:e])       # [H, pad64(n)]
                    ops.gemm_nt(dlt, xt, out=gw_tmp, accumulate=not first)
                    del dlt, xt
                first = False
        loss = torch.empty((), device=dev, dtype=torch.float32)
        ops.loss_reduce(row_loss, denom, loss)
        if need_grad:
            # the unscaled weight gradient sits in the gradient arena itself when the block was fresh (saved as None: a stage keeps no arena view)
            ctx.save_for_backward(dx, None if gw_tmp is blk.grad else gw_tmp, rows)
        ctx.meta = (arena, wkey, M_all)
        return loss

    @staticmethod
    def backward(ctx, g):
        dx, gw_tmp, rows = ctx.saved_tensors
        arena, wkey, M_all = ctx.meta
        inplace = gw_tmp is None
        blk = arena[wkey]
        g32 = g.reshape(1).float()
        ops.scale_add_(dx, dx, g32, accumulate=False)
        if rows is not None:
            dx = ops.scatter_rows(dx, rows, M_all)
        if inplace:
            ops.scale_add_(blk.grad, blk.grad, g32, accumulate=False)
        else:
            ops.scale_add_(gw_tmp, blk.grad, g32, accumulate=True)
        arena.grad_written(blk)
        return dx, None, None, None, None, None, None


# ---------------------------------------------------------------------------------------------- the stages as registered operators
# <Stage>Fn.apply(...) - what modeling.py calls - dispatches torch.ops.afk.<stage> (stage_ops.py): schema, fake implementation, backward operator
# with the gradient arena declared as mutated, autograd registration.  Slot kinds per forward argument: T tensor, T? optional tensor, A arena, S static.
from .stage_ops import register_stage  # noqa: E402

ConvStemFn.apply = staticmethod(register_stage("conv_stem", ConvStemFn, ("T", "T", "A", "S", "T", "S", "S", "S")))
EncoderLayerFn.apply = staticmethod(register_stage("encoder_layer", EncoderLayerFn, ("T", "T", "A", "S", "S", "S", "S", "T?")))
PoolNormFn.apply = staticmethod(register_stage("pool_norm", PoolNormFn, ("T", "T", "A", "S", "S", "S")))
RotaryTimeFn.apply = staticmethod(register_stage("rotary_time_stage", RotaryTimeFn, ("T", "T", "T", "A")))
ProjectorFn.apply = staticmethod(register_stage("projector", ProjectorFn, ("T", "T", "A", "S")))
EmbedScatterFn.apply = staticmethod(register_stage("embed_scatter", EmbedScatterFn, ("T?", "T", "A", "S", "T", "T?")))
DecoderLayerFn.apply = staticmethod(register_stage("decoder_layer", DecoderLayerFn, ("T", "T", "A", "S", "S", "S", "S", "S", "S", "S", "T", "T", "T?", "T?", "T?", "T?", "T?")))
RMSNormFn.apply = staticmethod(register_stage("final_rms_norm", RMSNormFn, ("T", "T", "A", "S", "S")))
LMHeadFn.apply = staticmethod(register_stage("lm_head", LMHeadFn, ("T", "T", "A", "S")))
LMHeadLossFn.apply = staticmethod(register_stage("lm_head_loss", LMHeadLossFn, ("T", "T", "A", "S", "T", "T", "T?")))
