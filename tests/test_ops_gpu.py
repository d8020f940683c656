"""Op-level parity of every HIP kernel against a plain PyTorch fp32 expression of the oracle op it replaces
(the oracle line is cited in include/afk.h).  All calls go through the C ABI (ops.py -> ctypes -> libafk.so)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _ops():
    from audio_flamingo_amd import ops

    return ops


def _cmp(name, got, ref, atol, rtol):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    if bad.any():
        idx = bad.nonzero()[:8].tolist()
        worst = err.argmax().item()
        raise AssertionError(
            f"{name}: {int(bad.sum())}/{bad.numel()} out of tol (atol={atol}, rtol={rtol}); max_err={err.max().item():.4g} "
            f"at flat {worst} got={got.flatten()[worst].item():.5g} ref={ref.flatten()[worst].item():.5g}; first bad idx={idx}; "
            f"ref_absmax={ref.abs().max().item():.4g}"
        )


def _rand(shape, dev, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (300, 260, 192), (1000, 1280, 1280), (129, 132, 64)])
def test_gemm_plain(dev, M, N, K):
    ops = _ops()
    a = _rand((M, K), dev, seed=1).to(BF)
    b = _rand((N, K), dev, seed=2).to(BF)  # asymmetric operands: catches row/col swaps
    c = ops.gemm_nt(a, b)
    ref = a.float() @ b.float().t()
    _cmp(f"gemm {M}x{N}x{K}", c, ref, atol=0.02 * math.sqrt(K), rtol=1e-2)


def test_default_build_has_no_probe_paths(dev):
    """the shipped libafk.so refuses every probe selector of afk_gemm_set_variant (wrong-result timing probes, rejected schedules) and
    ignores AFK_ATTN_DBG: a stray value cannot corrupt results"""
    from audio_flamingo_amd import _lib
    if _lib.has_probes():
        pytest.skip("probe build")
    ops = _ops()
    for bad in (3, 6, 10, 13, 2 + 256 * 0x40, 2 + 256 * 0x80):
        with pytest.raises(_lib.AfkError, match="AFK_PROBES"):
            ops.gemm_set_variant(bad)
    ops.gemm_set_variant(2 + 256 * 8)   # rasterization group height stays a legal knob
    ops.gemm_set_variant(0)


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 256, 128), (512, 768, 192), (300, 260, 320), (1000, 1280, 1280), (8, 512, 4096),
                                   (777, 1028, 64), (2048, 256, 2048)])
def test_gemm_variants(dev, variant, M, N, K):
    """both NT kernels (1: 128x128; 2: 256x256 8-wave ping-pong) on every edge shape: K-tiles 1/2/3/many (prologue + tail waits), M/N tails,
    tiny M.  (The rejected 256x256 schedules and their tests live in tools/probes/ and need a `make PROBES=1` build.)"""
    ops = _ops()
    ops.gemm_set_variant(variant)
    try:
        a = _rand((M, K), dev, seed=11).to(BF)
        b = _rand((N, K), dev, seed=12).to(BF)
        bias = _rand((N,), dev, seed=13).to(BF)
        c = ops.gemm_nt(a, b, bias=bias)
        ref = a.float() @ b.float().t() + bias.float()
        _cmp(f"gemm v{variant} {M}x{N}x{K}", c, ref, atol=0.02 * math.sqrt(K), rtol=1e-2)
        # repeat to shake out races between the LDS-DMA ring and the fragment reads: results must be bit-identical
        for _ in range(3):
            assert torch.equal(ops.gemm_nt(a, b, bias=bias), c), "non-deterministic GEMM result (LDS race?)"
    finally:
        ops.gemm_set_variant(0)


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 768, 192), (304, 264, 320), (1000, 1280, 1280), (8, 512, 4096), (2048, 256, 2048)])
def test_gemm_nn_dgrad_form(dev, M, N, K):
    """C = A[M,K] . Bt[K,N]  (B stored reduction-major, as dX = dY . W needs it)"""
    ops = _ops()
    a = _rand((M, K), dev, seed=21).to(BF)
    bt = _rand((K, N), dev, seed=22).to(BF)
    c = ops.gemm(a, bt, trans_b=True)
    _cmp(f"gemm NN {M}x{N}x{K}", c, a.float() @ bt.float(), atol=0.02 * math.sqrt(K), rtol=1e-2)
    for _ in range(3):
        assert torch.equal(ops.gemm(a, bt, trans_b=True), c), "non-deterministic NN GEMM (LDS race?)"


@pytest.mark.parametrize("form", ["NT", "NN", "TN"])
def test_gemm_generic_epilogue_instantiations(dev, form):
    """the runtime-flag instantiations (<.., -1>: narrow 8-byte stores because N % 8 != 0, bias + residual + accumulate epilogues on the
    transposed-operand kernels) - one out-of-line epilogue function per kernel since round 4 (they spilled 730 VGPRs inlined) - are counted
    (`gemm_generic_epilogue`) and give the same values as fp32"""
    ops = _ops()
    # NT: N % 8 = 4 -> no 16-byte epilogue -> generic instantiation.  NN / TN need N % 8 == 0 (16-byte rows of the reduction-major operand): there the
    # bias + residual epilogue - not one of the step's instantiations {plain, accumulate, split-K partials} - selects it
    M, N, K = (520, 260, 320) if form == "NT" else (520, 264, 320)
    a = _rand((K, M) if form == "TN" else (M, K), dev, seed=51).to(BF)
    b = _rand((N, K) if form == "NT" else (K, N), dev, seed=52).to(BF)
    bias = _rand((N,), dev, seed=53).to(BF)
    res = _rand((M, N), dev, seed=54).to(BF)
    af = a.float().t() if form == "TN" else a.float()
    bf = b.float().t() if form == "NT" else b.float()
    ops.gemm_set_variant(2)      # the 256x256 kernels also for this small NT shape
    try:
        ops.kernel_counts(reset=True)
        if form == "NT":
            c = ops.gemm_nt(a, b, bias=bias, residual=res)
        else:
            c = ops.gemm(a, b, trans_a=form == "TN", trans_b=True, bias=bias, residual=res, _splits=1)
        assert ops.kernel_counts()["gemm_generic_epilogue"] == 1
        ref = (af @ bf + bias.float()).to(BF).float() + res.float()
        _cmp(f"generic epilogue {form}", c, ref, atol=0.02 * math.sqrt(K), rtol=1e-2)
        if form != "NT":
            acc = ops.gemm(a, b, out=res.clone(), trans_a=form == "TN", trans_b=True, accumulate=True, _splits=1)
            _cmp(f"generic accumulate {form}", acc, af @ bf + res.float(), atol=0.02 * math.sqrt(K), rtol=1e-2)
        # a wide shape with the step's flags must NOT land there
        ops.kernel_counts(reset=True)
        a2 = _rand((K, 512) if form == "TN" else (512, K), dev, seed=55).to(BF)
        b2 = _rand((256, K) if form == "NT" else (K, 256), dev, seed=56).to(BF)
        ops.gemm_nt(a2, b2) if form == "NT" else ops.gemm(a2, b2, trans_a=form == "TN", trans_b=True, _splits=1)
        assert ops.kernel_counts()["gemm_generic_epilogue"] == 0
    finally:
        ops.gemm_set_variant(0)


@pytest.mark.parametrize("M,N,K,splits", [(512, 768, 4096, 4), (304, 264, 8192, 16), (1000, 1280, 4096 + 192, 3), (2048, 256, 2048, 2)])
def test_gemm_nn_splitk(dev, M, N, K, splits):
    """the NN form with an explicit split-K plan (the lm_head dgrad's path): fp32 partials folded in fixed order, ragged last split, bias epilogue"""
    ops = _ops()
    a = _rand((M, K), dev, seed=41).to(BF)
    bt = _rand((K, N), dev, seed=42).to(BF)
    bias = _rand((N,), dev, seed=43).to(BF)
    c = ops.gemm(a, bt, trans_b=True, bias=bias, _splits=splits)
    _cmp(f"gemm NN split-K {M}x{N}x{K}/{splits}", c, a.float() @ bt.float() + bias.float(), atol=0.02 * math.sqrt(K), rtol=1e-2)
    one = ops.gemm(a, bt, trans_b=True, bias=bias, _splits=1)
    _cmp("split-K vs one pass", c, one.float(), atol=0.02 * math.sqrt(K) / 8, rtol=2e-2)
    for _ in range(3):
        assert torch.equal(ops.gemm(a, bt, trans_b=True, bias=bias, _splits=splits), c), "non-deterministic NN split-K GEMM"


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 768, 192), (304, 264, 320), (1280, 1280, 12000), (3840, 1280, 6000), (8, 512, 100),
                                   (384, 1280, 24000), (2048, 256, 2048)])
def test_gemm_tn_wgrad_form(dev, M, N, K):
    """C = At[K,M]^T . Bt[K,N]  (both reduction-major; ragged K exercises the in-kernel tail mask)"""
    ops = _ops()
    at = _rand((K, M), dev, seed=31).to(BF)
    bt = _rand((K, N), dev, seed=32).to(BF)
    base = _rand((M, N), dev, seed=33).to(BF)
    c = ops.gemm(at, bt, trans_a=True, trans_b=True)
    ref = at.float().t() @ bt.float()
    _cmp(f"gemm TN {M}x{N}x{K}", c, ref, atol=0.02 * math.sqrt(K), rtol=1e-2)
    c2 = base.clone()
    ops.gemm(at, bt, out=c2, trans_a=True, trans_b=True, accumulate=True)
    _cmp("gemm TN accumulate", c2, ref + base.float(), atol=0.03 * math.sqrt(K), rtol=1e-2)
    for _ in range(3):
        assert torch.equal(ops.gemm(at, bt, trans_a=True, trans_b=True), c), "non-deterministic TN GEMM (LDS race?)"


def test_colsum(dev):
    ops = _ops()
    for rows, cols in [(100, 64), (12000, 1280), (8192, 4608), (513, 72)]:
        x = _rand((rows, cols), dev, seed=rows).to(BF)
        out = torch.ones(cols, device=dev, dtype=BF)
        ops.colsum(x, out, accumulate=True)
        _cmp("colsum", out, x.float().sum(0) + 1.0, atol=0.02 * math.sqrt(rows) + 0.05, rtol=1e-2)
        # one launch (the last row slice of a column block folds, round 4) == partial + fold launches, bit for bit, call after call on the same counters
        assert ops.COLSUM_FUSED
        try:
            ops.COLSUM_FUSED = False
            two = ops.colsum(x, torch.ones(cols, device=dev, dtype=BF), accumulate=True)
        finally:
            ops.COLSUM_FUSED = True
        for _ in range(3):
            assert torch.equal(ops.colsum(x, torch.ones(cols, device=dev, dtype=BF), accumulate=True), two)
        assert int(ops._colsum_counters(dev, 1).abs().sum()) == 0


def test_gemm_identity_layout(dev):
    """A = I pattern with asymmetric B: output must equal B^T rows exactly (detects transposed C writes)."""
    ops = _ops()
    M = N = 128
    K = 128
    a = torch.eye(M, K, device=dev).to(BF)
    b = (torch.arange(N * K, device=dev).reshape(N, K) % 251).float().to(BF)
    c = ops.gemm_nt(a, b)
    ref = b.float().t()[:M, :N]
    _cmp("gemm identity", c, ref, atol=0, rtol=0)


def test_gemm_epilogues(dev):
    ops = _ops()
    M, N, K = 384, 512, 256
    a = _rand((M, K), dev, 0.5, 3).to(BF)
    b = _rand((N, K), dev, 0.1, 4).to(BF)
    bias = _rand((N,), dev, 1.0, 5).to(BF)
    res = _rand((M, N), dev, 1.0, 6).to(BF)
    lin = (a.float() @ b.float().t() + bias.float()).to(BF).float()
    # bias + gelu (+preact)
    pre = torch.empty((M, N), device=dev, dtype=BF)
    c = ops.gemm_nt(a, b, bias=bias, gelu=True, preact_out=pre)
    _cmp("preact", pre, lin, atol=2e-2, rtol=1e-2)
    _cmp("bias+gelu", c, torch.nn.functional.gelu(lin), atol=2e-2, rtol=1e-2)
    # bias + residual
    c = ops.gemm_nt(a, b, bias=bias, residual=res)
    _cmp("bias+res", c, lin + res.float(), atol=3e-2, rtol=1e-2)
    # residual table broadcast (row m % 96)
    tab = _rand((96, N), dev, 1.0, 7).to(BF)
    c = ops.gemm_nt(a, b, residual=tab, res_mod=96)
    ref = (a.float() @ b.float().t()).to(BF).float() + tab.float().repeat(M // 96, 1)
    _cmp("res_mod", c, ref, atol=3e-2, rtol=1e-2)
    # f32 out + accumulate
    c32 = torch.ones((M, N), device=dev, dtype=torch.float32)
    ops.gemm_nt(a, b, out=c32, accumulate=True)
    _cmp("f32 accum", c32, a.float() @ b.float().t() + 1.0, atol=1e-2, rtol=1e-3)
    # bf16 accumulate with strided operands (views into wider buffers)
    wide_a = _rand((M, K + 64), dev, 0.5, 8).to(BF)
    cb = res.clone()
    ops.gemm_nt(wide_a[:, 64:], b, out=cb, accumulate=True)
    _cmp("bf16 accum strided", cb, wide_a[:, 64:].float() @ b.float().t() + res.float(), atol=4e-2, rtol=1e-2)


def test_transpose(dev):
    ops = _ops()
    for R, C in [(64, 64), (100, 72), (1500, 128), (8, 200)]:
        x = _rand((R, C), dev, seed=R).to(BF)
        t = ops.transpose(x)
        rp = (R + 63) // 64 * 64
        assert t.shape == (C, rp)
        assert torch.equal(t[:, :R], x.t()), f"transpose {R}x{C} mismatch"
        assert (t[:, R:] == 0).all(), "transpose pad not zero"
    # heads
    B, S, H, D = 2, 100, 3, 64
    ld = H * D + 32
    buf = _rand((B * S, ld), dev, seed=9).to(BF)
    t = ops.transpose_heads(buf, B, S, H, D, ld, 128)
    ref = buf[:, : H * D].reshape(B, S, H, D).permute(0, 2, 3, 1)
    assert torch.equal(t[..., :S], ref) and (t[..., S:] == 0).all()


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("rows,D", [(37, 128), (1000, 1280), (64, 3584)])
def test_layernorm(dev, rows, D):
    ops = _ops()
    x = _rand((rows, D), dev, 2.0, 1).to(BF)
    w = (1 + 0.1 * _rand((D,), dev, seed=2)).to(BF)
    b = (0.1 * _rand((D,), dev, seed=3)).to(BF)
    y, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-5)
    xr = x.float().requires_grad_(True)
    wr, br = w.float().requires_grad_(True), b.float().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), wr, br, 1e-5)
    _cmp("ln fwd", y, ref, atol=2e-2, rtol=1e-2)
    dy = _rand((rows, D), dev, 1.0, 4).to(BF)
    ref.backward(dy.float())
    dw = torch.empty(D, device=dev, dtype=BF)
    db = torch.empty(D, device=dev, dtype=BF)
    dx = ops.layernorm_bwd(x, w, dy, mean, rstd, dw, db)
    _cmp("ln dx", dx, xr.grad, atol=3e-2, rtol=2e-2)
    _cmp("ln dw", dw, wr.grad, atol=0.02 * math.sqrt(rows) + 0.05, rtol=2e-2)
    _cmp("ln db", db, br.grad, atol=0.02 * math.sqrt(rows) + 0.05, rtol=2e-2)
    # fused residual merge + accumulate
    skip = _rand((rows, D), dev, 1.0, 5).to(BF)
    dw2, db2 = dw.clone(), db.clone()
    dx2 = ops.layernorm_bwd(x, w, dy, mean, rstd, dw2, db2, dx_add=skip, accumulate=True)
    _cmp("ln dx+skip", dx2, xr.grad + skip.float(), atol=4e-2, rtol=2e-2)
    _cmp("ln dw acc", dw2, 2 * wr.grad, atol=0.05 * math.sqrt(rows) + 0.1, rtol=3e-2)


@pytest.mark.parametrize("rows,D", [(37, 96), (512, 3584)])
def test_rmsnorm(dev, rows, D):
    ops = _ops()
    x = _rand((rows, D), dev, 2.0, 1).to(BF)
    w = (1 + 0.1 * _rand((D,), dev, seed=2)).to(BF)
    y, rstd = ops.rmsnorm_fwd(x, w, 1e-6)

    def oracle(xf, wf):  # Qwen2RMSNorm.forward restated
        var = xf.pow(2).mean(-1, keepdim=True)
        return wf * (xf * torch.rsqrt(var + 1e-6))

    xr, wr = x.float().requires_grad_(True), w.float().requires_grad_(True)
    ref = oracle(xr, wr)
    _cmp("rms fwd", y, ref, atol=2e-2, rtol=1e-2)
    dy = _rand((rows, D), dev, 1.0, 4).to(BF)
    ref.backward(dy.float())
    dw = torch.empty(D, device=dev, dtype=BF)
    dx = ops.rmsnorm_bwd(x, w, dy, rstd, dw)
    _cmp("rms dx", dx, xr.grad, atol=3e-2, rtol=2e-2)
    _cmp("rms dw", dw, wr.grad, atol=0.02 * math.sqrt(rows) + 0.05, rtol=2e-2)


@pytest.mark.parametrize("kind,rows,D", [("rms", 8192, 3584), ("ln", 12000, 1280), ("rms", 4099, 3584), ("ln", 3001, 1280), ("rms", 70, 100), ("ln", 70, 100)])
def test_norm_bwd_training_shapes(dev, kind, rows, D):
    """the backward kernels at the shapes the training step launches them with (decoder RMSNorm 8192 x 3584, encoder LayerNorm 12000 x 1280:
    several row groups per block, ragged last group), with the fused residual-gradient merge and gradient accumulation, against fp32 torch;
    D = 100 (D % 8 != 0) takes the row-per-wave form.  Repeated launches are bit-identical (fixed-order reductions)."""
    ops = _ops()
    x = _rand((rows, D), dev, 2.0, 1).to(BF)
    w = (1 + 0.1 * _rand((D,), dev, seed=2)).to(BF)
    b = (0.1 * _rand((D,), dev, seed=3)).to(BF)
    dy = _rand((rows, D), dev, 1.0, 4).to(BF)
    skip = _rand((rows, D), dev, 1.0, 5).to(BF)
    xr, wr, br = x.float().requires_grad_(True), w.float().requires_grad_(True), b.float().requires_grad_(True)
    if kind == "ln":
        y, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-5)
        torch.nn.functional.layer_norm(xr, (D,), wr, br, 1e-5).backward(dy.float())
    else:
        y, rstd = ops.rmsnorm_fwd(x, w, 1e-6)
        (wr * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6))).backward(dy.float())
    base = (0.5 * _rand((D,), dev, seed=6)).to(BF)

    def run():
        dw, db = base.clone(), base.clone()
        if kind == "ln":
            dx = ops.layernorm_bwd(x, w, dy, mean, rstd, dw, db, dx_add=skip, accumulate=True)
        else:
            dx = ops.rmsnorm_bwd(x, w, dy, rstd, dw, dx_add=skip, accumulate=True)
        return dx, dw, db

    dx, dw, db = run()
    _cmp(f"{kind} dx+skip", dx, xr.grad + skip.float(), atol=4e-2, rtol=2e-2)
    _cmp(f"{kind} dw acc", dw, wr.grad + base.float(), atol=0.02 * math.sqrt(rows) + 0.05, rtol=2e-2)
    if kind == "ln":
        _cmp("ln db acc", db, br.grad + base.float(), atol=0.02 * math.sqrt(rows) + 0.05, rtol=2e-2)
    # the column sums are sharp too: relative L2 error of the whole dw vector well under a percent
    rel = float((dw.float() - (wr.grad + base.float())).norm() / (wr.grad + base.float()).norm())
    assert rel < 5e-3, rel
    for _ in range(2):
        dx2, dw2, db2 = run()
        assert torch.equal(dx, dx2) and torch.equal(dw, dw2) and torch.equal(db, db2), "norm backward not bit-deterministic"


# ------------------------------------------------------------------------------------------------ elementwise
def test_gelu_silu_rope_misc(dev):
    ops = _ops()
    x = _rand((64, 512), dev, 2.0, 1).to(BF)
    _cmp("gelu fwd", ops.gelu_fwd(x), torch.nn.functional.gelu(x.float()), 1e-2, 1e-2)
    dy = _rand((64, 512), dev, 1.0, 2).to(BF)
    xr = x.float().requires_grad_(True)
    torch.nn.functional.gelu(xr).backward(dy.float())
    _cmp("gelu bwd", ops.gelu_bwd(dy, x), xr.grad, 1e-2, 1e-2)
    # swiglu
    I = 256
    gu = _rand((40, 2 * I), dev, 1.5, 3).to(BF)
    gr = gu.float().requires_grad_(True)
    ref = torch.nn.functional.silu(gr[:, :I]) * gr[:, I:]
    h = ops.silu_mul_fwd(gu)
    _cmp("silu_mul fwd", h, ref, 2e-2, 1e-2)
    dh = _rand((40, I), dev, 1.0, 4).to(BF)
    ref.backward(dh.float())
    _cmp("silu_mul bwd", ops.silu_mul_bwd(gu, dh), gr.grad, 2e-2, 2e-2)
    # rope: oracle apply_rotary_pos_emb
    B, S, Hq, Hkv, D = 2, 24, 4, 2, 64
    ld = (Hq + 2 * Hkv) * D
    buf = _rand((B * S, ld), dev, 1.0, 5).to(BF)
    inv = 1.0 / (1e6 ** (torch.arange(0, D, 2, device=dev).float() / D))
    fr = torch.arange(S, device=dev).float()[:, None] * inv[None]
    emb = torch.cat([fr, fr], -1)
    cos, sin = emb.cos().to(BF), emb.sin().to(BF)

    def rot(t):
        return torch.cat([-t[..., D // 2:], t[..., : D // 2]], -1)

    qk = buf[:, : (Hq + Hkv) * D].float().reshape(B, S, Hq + Hkv, D)
    ref = qk * cos.float()[None, :, None] + rot(qk) * sin.float()[None, :, None]
    out = buf.clone()
    ops.rope_(out, cos, sin, S=S, nheads=Hq + Hkv, D=D)
    _cmp("rope fwd", out[:, : (Hq + Hkv) * D].reshape(B, S, Hq + Hkv, D), ref, 2e-2, 1e-2)
    assert torch.equal(out[:, (Hq + Hkv) * D:], buf[:, (Hq + Hkv) * D:]), "rope touched v columns"
    # backward = transpose of the rotation: <rope(x), y> == <x, rope_bwd(y)>
    y = _rand((B * S, ld), dev, 1.0, 6).to(BF)
    yb = y.clone()
    ops.rope_(yb, cos, sin, S=S, nheads=Hq + Hkv, D=D, backward=True)
    n = (Hq + Hkv) * D
    lhs = (out[:, :n].float() * y[:, :n].float()).sum()
    rhs = (buf[:, :n].float() * yb[:, :n].float()).sum()
    assert abs(lhs - rhs) < 2e-2 * abs(lhs) + 1.0, f"rope adjoint {lhs} vs {rhs}"
    # add / cast / rowsum
    a, b = _rand((33, 64), dev, seed=7).to(BF), _rand((33, 64), dev, seed=8).to(BF)
    _cmp("add", ops.add(a, b), a.float() + b.float(), 1e-2, 1e-2)
    f = _rand((1000,), dev, seed=9)
    assert torch.equal(ops.cast_f32_bf16(f), f.to(BF))
    xt = _rand((50, 192), dev, seed=10).to(BF)
    out = torch.zeros(50, device=dev, dtype=BF)
    ops.rowsum(xt, 150, out)
    _cmp("rowsum", out, xt[:, :150].float().sum(1), 0.1, 1e-2)


def test_conv_stem_helpers(dev):
    ops = _ops()
    W, C, T, Co = 2, 16, 100, 32
    x = _rand((W, C, T), dev, 1.0, 1)
    w1 = _rand((Co, C, 3), dev, 0.2, 2).to(BF)
    for xin in (x, x.to(BF)):
        col = ops.im2col_conv1(xin.contiguous())
        wp = ops.conv_weight_to_gemm(w1)
        y = ops.gemm_nt(torch.nn.functional.pad(col, (0, 64 - col.shape[1] % 64)) if col.shape[1] % 64 else col,
                        torch.nn.functional.pad(wp, (0, 64 - wp.shape[1] % 64)) if wp.shape[1] % 64 else wp)
        ref = torch.nn.functional.conv1d(xin.to(BF).float(), w1.float(), padding=1)  # [W, Co, T]
        _cmp("conv1 via im2col", y.reshape(W, T, Co).permute(0, 2, 1), ref, 3e-2, 2e-2)
    # conv2 (stride 2) from time-major input
    Ci = 64
    h = _rand((W * T, Ci), dev, 1.0, 3).to(BF)
    w2 = _rand((Co, Ci, 3), dev, 0.1, 4).to(BF)
    col = ops.im2col_conv2(h, W, T, Ci)
    y = ops.gemm_nt(col, ops.conv_weight_to_gemm(w2))
    ref = torch.nn.functional.conv1d(h.float().reshape(W, T, Ci).permute(0, 2, 1), w2.float(), stride=2, padding=1)
    _cmp("conv2 via im2col", y.reshape(W, T // 2, Co).permute(0, 2, 1), ref, 5e-2, 2e-2)
    # col2im is the adjoint of im2col
    dcol = _rand(tuple(col.shape), dev, 1.0, 5).to(BF)
    dh = ops.col2im_conv2(dcol, W, T, Ci)
    lhs = (col.float() * dcol.float()).sum()
    rhs = (h.float() * dh.float()).sum()
    assert abs(lhs - rhs) < 2e-2 * abs(lhs) + 2.0, f"col2im adjoint {lhs} vs {rhs}"
    # weight grad permute round trip
    dwp = _rand((Co, 3 * Ci), dev, 1.0, 6).to(BF)
    dw = torch.zeros((Co, Ci, 3), device=dev, dtype=BF)
    ops.conv_weight_grad_from_gemm(dwp, dw)
    assert torch.equal(dw, dwp.reshape(Co, 3, Ci).permute(0, 2, 1))
    # avgpool
    xp = _rand((40, 64), dev, 1.0, 7).to(BF)
    _cmp("pool fwd", ops.avgpool2_fwd(xp, 20, 64), xp.float().reshape(20, 2, 64).mean(1), 1e-2, 1e-2)
    dyp = _rand((20, 64), dev, 1.0, 8).to(BF)
    _cmp("pool bwd", ops.avgpool2_bwd(dyp, 20, 64), (0.5 * dyp.float()).repeat_interleave(2, 0), 1e-2, 1e-2)


def test_embed_scatter(dev):
    ops = _ops()
    V, H, B, S, AID = 50, 64, 3, 700, 49
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, V - 1, (B, S), generator=g)
    ids[0, 5:300] = AID
    ids[2, 600:700] = AID
    ids = ids.to(dev)
    n_audio = int((ids == AID).sum())
    embed = _rand((V, H), dev, 1.0, 1).to(BF)
    audio = _rand((n_audio, H), dev, 1.0, 2).to(BF)
    src, cnt = ops.placeholder_scan(ids, AID)
    assert int(cnt.item()) == n_audio
    out = ops.embed_scatter_fwd(ids.reshape(-1), src, embed, audio)
    ref = embed[ids.reshape(-1)].clone()
    ref[(ids == AID).reshape(-1)] = audio  # masked_scatter row-major order
    assert torch.equal(out, ref), "embed scatter mismatch"
    dout = _rand((B * S, H), dev, 1.0, 3).to(BF)
    d_embed = torch.zeros((V, H), device=dev, dtype=torch.bfloat16)
    d_audio = torch.empty((n_audio, H), device=dev, dtype=BF)
    ops.embed_scatter_bwd(ids.reshape(-1), src, dout, d_embed, d_audio)
    mask = (ids == AID).reshape(-1)
    assert torch.equal(d_audio, dout[mask])
    ref_de = torch.zeros((V, H), device=dev)
    ref_de.index_add_(0, ids.reshape(-1)[~mask], dout[~mask].float())
    # segmented fp32 accumulation in row order, ONE bf16 rounding: equal to the fp32 reference rounded once, and bit-deterministic
    assert torch.equal(d_embed, ref_de.to(BF)), float((d_embed.float() - ref_de).abs().max())
    for _ in range(3):
        d2 = torch.zeros_like(d_embed)
        ops.embed_scatter_bwd(ids.reshape(-1), src, dout, d2, None)
        assert torch.equal(d2, d_embed), "embed_tokens gradient is not bit-deterministic"
    # accumulate into an existing gradient (second micro-batch)
    ops.embed_scatter_bwd(ids.reshape(-1), src, dout, d2, None)
    _cmp("d_embed accumulate", d2, 2 * ref_de, atol=0.05, rtol=1e-2)
    # the same kernels as row gather / scatter of an activation matrix (lm_head on the rows with a label only)
    x = _rand((B * S, H), dev, 1.0, 4).to(BF)
    rows = torch.tensor(sorted(torch.randperm(B * S, generator=g)[:333].tolist()), device=dev)
    xs = ops.gather_rows(x, rows)
    assert torch.equal(xs, x[rows])
    back = ops.scatter_rows(xs, rows, B * S)
    ref_b = torch.zeros_like(x)
    ref_b[rows] = x[rows]
    assert torch.equal(back, ref_b)


def test_gemm_splitk(dev):
    """few output tiles + long reduction -> split-K path (fp32 partials + fixed-order reduce with the fused epilogue): values against
    fp32 torch, bit-determinism, and agreement with the one-pass kernel"""
    ops = _ops()
    for (M, N, K) in [(1, 3584, 18944), (8, 4608, 3584), (1280, 1280, 12032), (130, 260, 2048)]:
        assert ops.splitk_plan(M, N, K) > 1, (M, N, K)
        a = _rand((M, K), dev, 1.0, 1).to(BF)
        b = _rand((N, K), dev, 1.0, 2).to(BF)
        bias = _rand((N,), dev, 1.0, 3).to(BF)
        res = _rand((M, N), dev, 1.0, 4).to(BF)
        c = ops.gemm_nt(a, b, bias=bias, residual=res)
        ref = (a.float() @ b.float().T + bias.float()).to(BF).float() + res.float()
        _cmp(f"splitk {M}x{N}x{K}", c, ref, atol=K ** 0.5 * 2e-2, rtol=2e-2)
        for _ in range(2):
            assert torch.equal(ops.gemm_nt(a, b, bias=bias, residual=res), c), "split-K GEMM not deterministic"
        ops.SPLITK = False
        try:
            c1 = ops.gemm_nt(a, b, bias=bias, residual=res)
        finally:
            ops.SPLITK = True
        _cmp("splitk vs one-pass", c, c1.float(), atol=K ** 0.5 * 1e-2, rtol=2e-2)
        acc = c.clone()
        ops.gemm_nt(a, b, out=acc, accumulate=True)
        _cmp("splitk accumulate", acc, c.float() + (a.float() @ b.float().T), atol=K ** 0.5 * 3e-2, rtol=3e-2)


@pytest.mark.parametrize("M", [1, 2, 8])
def test_gemv_decode_shapes(dev, M):
    """skinny-M weight-streaming path (decode-time Linear layers): values vs fp32, bias + residual epilogue, ragged N / K, determinism"""
    ops = _ops()
    for (N, K) in [(3584, 3584), (4608, 3584), (1000, 1280), (3584, 18944), (72, 256)]:
        a = _rand((M, K), dev, 1.0, 1).to(BF)
        b = _rand((N, K), dev, 1.0, 2).to(BF)
        bias = _rand((N,), dev, 1.0, 3).to(BF)
        res = _rand((M, N), dev, 1.0, 4).to(BF)
        c = ops.gemm_nt(a, b, bias=bias, residual=res)
        ref = (a.float() @ b.float().T + bias.float()).to(BF).float() + res.float()
        _cmp(f"gemv {M}x{N}x{K}", c, ref, atol=K ** 0.5 * 2e-2, rtol=2e-2)
        assert torch.equal(ops.gemm_nt(a, b, bias=bias, residual=res), c)


def test_gemm_wide_epilogue_matches_narrow(dev):
    """16-byte epilogue (lane halves trade registers, 8 consecutive columns per lane) == the 8-byte form, bit for bit, for every fused
    epilogue; shapes with N % 8 != 0 or misaligned operands silently take the narrow form"""
    from audio_flamingo_amd import _lib
    ops = _ops()
    for (M, N, K) in [(300, 520, 256), (8192, 3584, 512), (1000, 1028, 128)]:
        a = _rand((M, K), dev, 1.0, 1).to(BF)
        b = _rand((N, K), dev, 1.0, 2).to(BF)
        bias = _rand((N,), dev, 1.0, 3).to(BF)
        res = _rand((M, N), dev, 1.0, 4).to(BF)
        outs = []
        for narrow in (0, 16):
            _lib.call("afk_gemm_set_variant", narrow)
            pre = torch.empty((M, N), device=dev, dtype=BF)
            y1 = ops.gemm_nt(a, b, bias=bias, gelu=True, preact_out=pre, residual=res)
            y2 = ops.gemm_nt(a, b, out_f32=True)
            y3 = res.clone()
            ops.gemm_nt(a, b, out=y3, accumulate=True)
            y4 = ops.gemm(a.t().contiguous(), b.t().contiguous(), trans_a=True, trans_b=True) if M % 8 == 0 and N % 8 == 0 else None
            outs.append((y1, pre, y2, y3, y4))
        _lib.call("afk_gemm_set_variant", 0)
        for x, y in zip(*outs):
            assert (x is None and y is None) or torch.equal(x, y)


def test_gemm_swiglu_bwd_epilogue(dev):
    """down-projection dgrad with the SwiGLU backward fused into its epilogue == dgrad GEMM followed by silu_mul_bwd, bit for bit"""
    ops = _ops()
    M, I, H = 700, 1024, 512
    dy = _rand((M, H), dev, 1.0, 1).to(BF)
    wt = _rand((I, H), dev, 0.1, 2).to(BF)       # W_down^T shadow: [I, H]
    gu = _rand((M, 2 * I), dev, 1.5, 3).to(BF)
    da = ops.gemm_nt(dy, wt)
    ref = ops.silu_mul_bwd(gu, da)
    got = ops.gemm_nt(dy, wt, swiglu_bwd=gu)
    assert got.shape == ref.shape and torch.equal(got, ref)


@pytest.mark.parametrize("M,I,K", [(700, 512, 256), (8192, 1024, 192), (1000, 128, 64), (2048, 18944, 128)])
def test_gemm_swiglu_fwd_epilogue(dev, M, I, K):
    """gate|up GEMM with the SwiGLU forward in its epilogue (paired gate / up tiles per workgroup) == GEMM followed by silu_mul_fwd, bit for
    bit, on both outputs; ragged M, one and many N tiles, the AF3 intermediate size"""
    ops = _ops()
    x = _rand((M, K), dev, 1.0, 1).to(BF)
    w = _rand((2 * I, K), dev, 0.5, 2).to(BF)
    gu_ref = ops.gemm_nt(x, w)
    h_ref = ops.silu_mul_fwd(gu_ref)
    h = torch.full((M, I), float("nan"), device=dev, dtype=BF)
    gu = ops.gemm_nt(x, w, swiglu_fwd_out=h)
    assert torch.equal(gu, gu_ref), "gate|up pre-activations differ"
    assert torch.equal(h, h_ref), "fused silu(gate) * up differs from the two-kernel form"
    assert torch.equal(ops.gemm_nt(x, w, swiglu_fwd_out=torch.empty_like(h)), gu)


def test_gemm_swiglu_fwd_refuses_unsupported_forms(dev):
    """the fused SwiGLU forward exists in ONE kernel instantiation; any dispatch state that cannot reach it (narrow-epilogue A/B
    knob, forced 128x128 variant) must raise - never return with swiglu_fwd_out unwritten - and variant 2 / auto must still serve it"""
    from audio_flamingo_amd import _lib
    ops = _ops()
    x = _rand((512, 128), dev, 1.0, 1).to(BF)
    w = _rand((512, 128), dev, 0.5, 2).to(BF)
    ref_h = ops.silu_mul_fwd(ops.gemm_nt(x, w))
    try:
        ops.gemm_set_variant(16)                      # 8-byte epilogue knob
        with pytest.raises(_lib.AfkError, match="SWIGLU_FWD"):
            ops.gemm_nt(x, w, swiglu_fwd_out=torch.empty((512, 256), device=dev, dtype=BF))
        for v in (0, 1, 2):                           # a forced small-tile variant still routes the flag to the ping-pong kernel
            ops.gemm_set_variant(v)
            h = torch.full((512, 256), float("nan"), device=dev, dtype=BF)
            ops.gemm_nt(x, w, swiglu_fwd_out=h)
            assert torch.equal(h, ref_h), v
    finally:
        ops.gemm_set_variant(0)


def test_gemm_tn_splitk(dev):
    """narrow weight gradients (encoder: 1280 x 1280 from 12 000 rows) on the TN kernel with split-K: values, determinism, accumulate"""
    ops = _ops()
    for (M, N, K) in [(1280, 1280, 12000), (3840, 1280, 12000), (520, 264, 4100)]:
        assert ops.splitk_plan_256(M, N, K) > 1
        at = _rand((K, M), dev, 1.0, 1).to(BF)
        bt = _rand((K, N), dev, 1.0, 2).to(BF)
        c = ops.gemm(at, bt, trans_a=True, trans_b=True)
        ref = at.float().T @ bt.float()
        _cmp(f"tn splitk {M}x{N}x{K}", c, ref, atol=K ** 0.5 * 2e-2, rtol=2e-2)
        assert torch.equal(ops.gemm(at, bt, trans_a=True, trans_b=True), c), "TN split-K not deterministic"
        acc = c.clone()
        ops.gemm(at, bt, out=acc, trans_a=True, trans_b=True, accumulate=True)
        _cmp("tn splitk accumulate", acc, 2 * ref, atol=K ** 0.5 * 4e-2, rtol=3e-2)


@pytest.mark.parametrize("M,N,K", [(8448, 2048, 4096), (2048, 8704, 4096)])
def test_gemm_tn_peeled_tail(dev, request, M, N, K):
    """TN weight gradients whose tile count ends in a nearly empty round (264 / 272 tiles on 256 CUs): the last tile rows / columns run as a
    split-K strip (ops.peel_plan_256).  Same values as the single launch outside the strip (bit for bit), fp32-reference tolerance inside,
    deterministic, accumulate honoured on both parts"""
    ops = _ops()
    plan = ops.peel_plan_256(M, N, K)
    assert plan is not None and plan[0] == (0 if M > N else 1), plan
    at = _rand((K, M), dev, 1.0, 1).to(BF)
    bt = _rand((K, N), dev, 1.0, 2).to(BF)
    one = ops.gemm(at, bt, trans_a=True, trans_b=True)
    was, ops.PEEL_TAIL = ops.PEEL_TAIL, True   # opt-in (AFK_PEEL_TAIL=1): slower on the overlapped step, see ops.py
    request.addfinalizer(lambda: setattr(ops, "PEEL_TAIL", was))
    ops.kernel_counts(reset=True)
    c = ops.gemm(at, bt, trans_a=True, trans_b=True)
    cnt = ops.kernel_counts()
    assert cnt["gemm_tn256"] == 2 and cnt["gemm_splitk"] == 1, cnt
    axis, cut, _ = plan
    main_c, main_1 = (c[:cut], one[:cut]) if axis == 0 else (c[:, :cut], one[:, :cut])
    assert torch.equal(main_c, main_1), "rows / columns outside the strip must not change"
    ref = at.float().T @ bt.float()
    _cmp("tn peeled", c, ref, atol=K ** 0.5 * 2e-2, rtol=2e-2)
    assert torch.equal(ops.gemm(at, bt, trans_a=True, trans_b=True), c), "peeled TN GEMM not deterministic"
    acc = c.clone()
    ops.gemm(at, bt, out=acc, trans_a=True, trans_b=True, accumulate=True)
    _cmp("tn peeled accumulate", acc, 2 * ref, atol=K ** 0.5 * 4e-2, rtol=3e-2)


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(qkv, B, S, Hq, Hkv, D, scale, causal, kv_len):
    q = qkv[:, : Hq * D].float().reshape(B, S, Hq, D).transpose(1, 2)
    k = qkv[:, Hq * D: (Hq + Hkv) * D].float().reshape(B, S, Hkv, D).transpose(1, 2)
    v = qkv[:, (Hq + Hkv) * D:].float().reshape(B, S, Hkv, D).transpose(1, 2)
    g = Hq // Hkv
    k, v = k.repeat_interleave(g, 1), v.repeat_interleave(g, 1)
    s = (q @ k.transpose(-1, -2)) * scale
    mask = torch.zeros(B, 1, S, S, dtype=torch.bool, device=qkv.device)
    if causal:
        mask |= torch.triu(torch.ones(S, S, dtype=torch.bool, device=qkv.device), 1)
    if kv_len is not None:
        mask |= (torch.arange(S, device=qkv.device)[None, :] >= kv_len[:, None].long())[:, None, None, :]
    s = s.masked_fill(mask, float("-inf"))
    p = torch.softmax(s, -1)
    o = (p @ v).transpose(1, 2).reshape(B * S, Hq * D)
    return o


@pytest.mark.parametrize("B,S,Hq,Hkv,D,causal,pad", [
    (2, 96, 4, 4, 64, False, False),
    (2, 150, 4, 4, 64, False, True),
    (2, 200, 4, 2, 128, True, False),
    (1, 77, 6, 2, 32, True, False),
    (1, 1500, 2, 2, 64, False, False),
])
def test_attention(dev, B, S, Hq, Hkv, D, causal, pad):
    ops = _ops()
    ld = (Hq + 2 * Hkv) * D
    qkv = _rand((B * S, ld), dev, 1.0, 1).to(BF)
    scale = D ** -0.5
    kv_len = None
    if pad:
        kv_len = torch.tensor([S, S - 37], device=dev, dtype=torch.int32)[:B]
    o, lse = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=scale, causal=causal, kv_len=kv_len)
    qr = qkv.float().requires_grad_(True)
    ref = _attn_ref(qr, B, S, Hq, Hkv, D, scale, causal, kv_len)
    _cmp("attn fwd", o, ref, atol=2e-2, rtol=2e-2)
    do = _rand((B * S, Hq * D), dev, 1.0, 2).to(BF)
    ref.backward(do.float())
    dqkv = ops.attn_bwd(qkv, o, do, lse, B, S, Hq, Hkv, D, scale=scale, causal=causal, kv_len=kv_len)
    nq, nk = Hq * D, Hkv * D
    _cmp("attn dq", dqkv[:, :nq], qr.grad[:, :nq], atol=3e-2, rtol=3e-2)
    _cmp("attn dk", dqkv[:, nq: nq + nk], qr.grad[:, nq: nq + nk], atol=4e-2, rtol=3e-2)
    _cmp("attn dv", dqkv[:, nq + nk:], qr.grad[:, nq + nk:], atol=4e-2, rtol=3e-2)


@pytest.mark.parametrize("B,S,Hq,Hkv,D,lo,hi", [
    (4, 200, 4, 2, 128, (0, 37, 64, 130), (200, 200, 190, 200)),     # lo inside a tile, on a tile edge, beyond the first query block
    (3, 330, 4, 4, 64, (5, 128, 300), (330, 300, 330)),
    (2, 1024, 14, 2, 128, (0, 700), (1024, 1024)),                     # GQA group 7 split-head sweep, whole key blocks of padding
])
def test_attention_left_padded(dev, B, S, Hq, Hkv, D, lo, hi):
    """causal attention over keys [kv_lo[b], kv_len[b]) - the left-padded batches of the reference processor - on the LDS-staged kernels:
    values vs the fp32 reference on the valid rows, exact zeros on the padded rows (output, dQ) and padded keys (dK, dV)"""
    ops = _ops()
    ld = (Hq + 2 * Hkv) * D
    qkv = _rand((B * S, ld), dev, 1.0, 11).to(BF)
    do = _rand((B * S, Hq * D), dev, 1.0, 12).to(BF)
    scale = D ** -0.5
    kv_lo = torch.tensor(lo, device=dev, dtype=torch.int32)
    kv_len = torch.tensor(hi, device=dev, dtype=torch.int32)
    ops.kernel_counts(reset=True)
    o, lse = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=scale, causal=True, kv_len=kv_len, kv_lo=kv_lo)
    dqkv = ops.attn_bwd(qkv, o, do, lse, B, S, Hq, Hkv, D, scale=scale, causal=True, kv_len=kv_len, kv_lo=kv_lo)
    cnt = ops.kernel_counts()
    fam = "d128" if D == 128 else "d64"
    assert cnt[f"attn2_fwd_{fam}"] == 1 and cnt[f"attn2_bwd_{fam}"] == 1, cnt
    # fp32 reference
    qr = qkv.float().requires_grad_(True)
    q = qr[:, : Hq * D].reshape(B, S, Hq, D).transpose(1, 2)
    k = qr[:, Hq * D: (Hq + Hkv) * D].reshape(B, S, Hkv, D).transpose(1, 2).repeat_interleave(Hq // Hkv, 1)
    v = qr[:, (Hq + Hkv) * D:].reshape(B, S, Hkv, D).transpose(1, 2).repeat_interleave(Hq // Hkv, 1)
    ar = torch.arange(S, device=dev)
    dead = (ar[None, :] < kv_lo[:, None].long()) | (ar[None, :] >= kv_len[:, None].long())      # [B, S] keys
    mask = dead[:, None, None, :] | torch.triu(torch.ones(S, S, dtype=torch.bool, device=dev), 1)[None, None]
    s_ = (q @ k.transpose(-1, -2)) * scale
    p_ = torch.nan_to_num(torch.softmax(s_.masked_fill(mask, float("-inf")), -1), nan=0.0)   # fully masked rows -> zero rows
    ref = (p_ @ v).transpose(1, 2).reshape(B * S, Hq * D)
    ref.backward(do.float())
    row_pad = (ar[None, :] < kv_lo[:, None].long()).reshape(-1)      # padded QUERY rows (left padding); right-padded rows still see keys
    key_pad = dead.reshape(-1)
    nq, nk = Hq * D, Hkv * D
    assert torch.isfinite(o.float()).all() and torch.isfinite(dqkv.float()).all()
    assert (o[row_pad] == 0).all() and (dqkv[row_pad, :nq] == 0).all(), "padded query rows must be exact zeros"
    assert (dqkv[key_pad, nq:] == 0).all(), "padded keys must get zero dK / dV"
    _cmp("lpad fwd", o, ref, atol=2e-2, rtol=2e-2)
    _cmp("lpad dq", dqkv[:, :nq], qr.grad[:, :nq], atol=3e-2, rtol=3e-2)
    _cmp("lpad dk", dqkv[:, nq: nq + nk], qr.grad[:, nq: nq + nk], atol=4e-2, rtol=3e-2)
    _cmp("lpad dv", dqkv[:, nq + nk:], qr.grad[:, nq + nk:], atol=4e-2, rtol=3e-2)


def test_attention_random_padding_sweep(dev):
    """seeded sweep over ragged shapes and paddings of the causal LDS kernels: S not a multiple of 64 / 128, kv_lo and kv_len anywhere (incl.
    lo == hi: a fully padded sample, and lo in the last tile), GQA groups 1 / 2 / 7, both head sizes - forward and all three gradients"""
    import random

    ops = _ops()
    rng = random.Random(1234)
    for case in range(10):
        D = rng.choice((64, 128))
        Hkv = rng.choice((1, 2))
        Hq = Hkv * rng.choice((1, 2, 7))
        B = rng.randint(1, 3)
        S = rng.choice((70, 129, 200, 257, 333, 450))
        hi = [rng.randint(1, S) for _ in range(B)]
        lo = [rng.randint(0, h) for h in hi]
        if case == 0:
            lo[0] = hi[0]                         # a sample that is padding only
        kv_lo = torch.tensor(lo, device=dev, dtype=torch.int32)
        kv_len = torch.tensor(hi, device=dev, dtype=torch.int32)
        qkv = _rand((B * S, (Hq + 2 * Hkv) * D), dev, 1.0, 100 + case).to(BF)
        do = _rand((B * S, Hq * D), dev, 1.0, 200 + case).to(BF)
        scale = D ** -0.5
        o, lse = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=scale, causal=True, kv_len=kv_len, kv_lo=kv_lo)
        dqkv = ops.attn_bwd(qkv, o, do, lse, B, S, Hq, Hkv, D, scale=scale, causal=True, kv_len=kv_len, kv_lo=kv_lo)
        qr = qkv.float().requires_grad_(True)
        q = qr[:, : Hq * D].reshape(B, S, Hq, D).transpose(1, 2)
        k = qr[:, Hq * D: (Hq + Hkv) * D].reshape(B, S, Hkv, D).transpose(1, 2).repeat_interleave(Hq // Hkv, 1)
        v = qr[:, (Hq + Hkv) * D:].reshape(B, S, Hkv, D).transpose(1, 2).repeat_interleave(Hq // Hkv, 1)
        ar = torch.arange(S, device=dev)
        dead = (ar[None, :] < kv_lo[:, None].long()) | (ar[None, :] >= kv_len[:, None].long())
        mask = dead[:, None, None, :] | torch.triu(torch.ones(S, S, dtype=torch.bool, device=dev), 1)[None, None]
        p_ = torch.nan_to_num(torch.softmax(((q @ k.transpose(-1, -2)) * scale).masked_fill(mask, float("-inf")), -1), nan=0.0)
        ref = (p_ @ v).transpose(1, 2).reshape(B * S, Hq * D)
        ref.backward(do.float())
        tag = f"sweep {case} (B={B} S={S} Hq={Hq} Hkv={Hkv} D={D} lo={lo} hi={hi})"
        assert torch.isfinite(o.float()).all() and torch.isfinite(dqkv.float()).all(), tag
        nq, nk = Hq * D, Hkv * D
        _cmp(tag + " fwd", o, ref, atol=2e-2, rtol=2e-2)
        _cmp(tag + " dq", dqkv[:, :nq], qr.grad[:, :nq], atol=3e-2, rtol=3e-2)
        _cmp(tag + " dk", dqkv[:, nq: nq + nk], qr.grad[:, nq: nq + nk], atol=4e-2, rtol=3e-2)
        _cmp(tag + " dv", dqkv[:, nq + nk:], qr.grad[:, nq + nk:], atol=4e-2, rtol=3e-2)


def test_attention_online_softmax_spike(dev):
    """force a late running-max jump (guide rule 26): one key row aligned with one query row at a late tile"""
    ops = _ops()
    B, S, H, D = 1, 160, 1, 64
    qkv = _rand((B * S, 3 * D), dev, 0.3, 3).to(BF)
    qkv[10, :D] = 4.0
    qkv[140, D: 2 * D] = 4.0
    o, _ = ops.attn_fwd(qkv, B, S, H, H, D, scale=D ** -0.5, causal=False)
    _cmp("attn spike", o, _attn_ref(qkv, B, S, H, H, D, D ** -0.5, False, None), 2e-2, 2e-2)


@pytest.mark.parametrize("name,B,S,Hq,Hkv,D,causal", [("encoder", 8, 1500, 20, 20, 64, False), ("decoder", 8, 1024, 28, 4, 128, True),
                                                     ("long", 1, 3000, 28, 4, 128, True)])
def test_attention_full_size_bit_deterministic(dev, name, B, S, Hq, Hkv, D, causal):
    """the AF3-7B attention shapes with the chip full (thousands of co-resident blocks): three runs must agree bit for bit (a missed
    LDS / MFMA hazard shows up as run-to-run noise only at this scale), and a sample of rows must match the fp32 reference"""
    ops = _ops()
    qkv = _rand((B * S, (Hq + 2 * Hkv) * D), dev, 1.0, 5).to(BF)
    do = _rand((B * S, Hq * D), dev, 1.0, 6).to(BF)
    scale = D ** -0.5
    outs = []
    for _ in range(3):
        o, lse = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=scale, causal=causal)
        dqkv = ops.attn_bwd(qkv, o, do, lse, B, S, Hq, Hkv, D, scale=scale, causal=causal)
        outs.append((o.clone(), lse.clone(), dqkv.clone()))
    torch.cuda.synchronize()
    for o, lse, dqkv in outs[1:]:
        assert torch.equal(o, outs[0][0]) and torch.equal(lse, outs[0][1]), f"{name}: forward not deterministic"
        assert torch.equal(dqkv, outs[0][2]), f"{name}: backward not deterministic"
    # fp32 reference on batch 0, first 2 query heads / their kv head (keeps the reference small)
    g = Hq // Hkv
    hq = min(2, g) if g > 1 else 2
    q = qkv[:S, : hq * D]
    k = qkv[:S, Hq * D: Hq * D + (D if g > 1 else hq * D)]
    v = qkv[:S, (Hq + Hkv) * D: (Hq + Hkv) * D + (D if g > 1 else hq * D)]
    sub = torch.cat([q, k, v], 1).contiguous()
    ref = _attn_ref(sub, 1, S, hq, 1 if g > 1 else hq, D, scale, causal, None)
    _cmp(f"{name} fwd sample", outs[0][0][:S, : hq * D], ref, atol=2e-2, rtol=2e-2)


def test_attention_lazy_rescale_ramp(dev):
    """scores that keep growing along the key axis, by small steps (below the rescale threshold) and by jumps (above it): the stale
    running max of the lazy-rescale forward must stay exact"""
    ops = _ops()
    B, S, H, D = 1, 1024, 2, 64
    qkv = _rand((B * S, 3 * H * D), dev, 0.05, 7).to(BF)
    ramp = torch.linspace(0.0, 6.0, S, device=dev)
    ramp[700:] += 8.0
    qkv[:, :D] = 1.0                                     # head 0 queries: all ones
    qkv[:, H * D: H * D + D] = (ramp[:, None] * torch.ones(D, device=dev)).to(BF)  # head 0 keys: k_j = ramp_j -> s_ij = D * ramp_j * scale
    o, lse = ops.attn_fwd(qkv, B, S, H, H, D, scale=D ** -0.5, causal=False)
    _cmp("ramp fwd", o, _attn_ref(qkv, B, S, H, H, D, D ** -0.5, False, None), 2e-2, 2e-2)
    oc, _ = ops.attn_fwd(qkv, B, S, H, H, D, scale=D ** -0.5, causal=True)
    _cmp("ramp fwd causal", oc, _attn_ref(qkv, B, S, H, H, D, D ** -0.5, True, None), 2e-2, 2e-2)


def test_rotary_time(dev):
    """Music Flamingo rotary time embedding kernel: forward vs the fp64 formula, backward = transposed rotation (checked through autograd)"""
    ops = _ops()
    rows, E, R = 300, 128, 52
    x = _rand((rows, E), dev, 1.0, 1).to(BF)
    ang = _rand((rows, R), dev, 3.0, 2)
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()

    def ref(xx):
        xx = xx.double()
        rot, rest = xx[:, :R], xx[:, R:]
        pr = rot.reshape(rows, -1, 2)
        half = torch.stack((-pr[..., 1], pr[..., 0]), -1).flatten(-2)
        return torch.cat((rot * cos.double() + half * sin.double(), rest), -1)

    y = ops.rotary_time(x, cos, sin)
    _cmp("rotary_time fwd", y, ref(x).float(), atol=1e-2, rtol=1e-2)
    assert torch.equal(y[:, R:], x[:, R:])
    xr = x.float().requires_grad_(True)
    dy = _rand((rows, E), dev, 1.0, 3).to(BF)
    ref(xr).backward(dy.double())
    _cmp("rotary_time bwd", ops.rotary_time(dy, cos, sin, backward=True), xr.grad, atol=1e-2, rtol=1e-2)


@pytest.mark.parametrize("D,Hq,Hkv", [(128, 8, 2), (64, 4, 4)])
def test_attn_decode_split_kv(dev, D, Hq, Hkv):
    """Q = 1 attention over a KV cache (split-KV kernel) against fp32 softmax attention; ragged visible intervals [lo, hi) per sample"""
    from audio_flamingo_amd import _lib
    ops = _ops()
    B, Smax = 3, 1000
    spad = ops.pad64(Smax)
    nk, nq = Hkv * D, Hq * D
    q = _rand((B, nq), dev, 1.0, 1).to(BF)
    kc = _rand((B, Smax, nk), dev, 1.0, 2).to(BF)
    v = _rand((B, Smax, nk), dev, 1.0, 3).to(BF)
    vt = torch.zeros((B, Hkv, D, spad), device=dev, dtype=BF)
    vt[..., :Smax] = v.reshape(B, Smax, Hkv, D).permute(0, 2, 3, 1)
    kr = torch.tensor([[0, 1000], [37, 801], [5, 6]], device=dev, dtype=torch.int32)
    for ns in (1, 8, 13):
        o = torch.empty((B, nq), device=dev, dtype=BF)
        ws = torch.empty(_lib.load().afk_attn_decode_workspace_floats(B, Hq, D, ns), device=dev, dtype=torch.float32)
        _lib.call("afk_attn_decode", q.data_ptr(), nq, D, kc.data_ptr(), Smax * nk, nk, D, vt.data_ptr(), Hkv * D * spad, spad, o.data_ptr(), nq, D,
                  kr.data_ptr(), B, Hq, Hkv, D, float(D ** -0.5), ns, ws.data_ptr(), ops._stream())
        g = Hq // Hkv
        for b in range(B):
            lo, hi = int(kr[b, 0]), int(kr[b, 1])
            qq = q[b].float().reshape(Hq, D)
            kk = kc[b, lo:hi].float().reshape(hi - lo, Hkv, D).repeat_interleave(g, 1)
            vv = v[b, lo:hi].float().reshape(hi - lo, Hkv, D).repeat_interleave(g, 1)
            p = torch.softmax(torch.einsum("hd,shd->hs", qq, kk) * D ** -0.5, -1)
            ref = torch.einsum("hs,shd->hd", p, vv).reshape(-1)
            _cmp(f"attn_decode ns={ns} b={b}", o[b], ref, atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("D,Hq,Hkv", [(128, 28, 4), (64, 4, 4)])
def test_attn_decode_fused_last_block_merges(dev, D, Hq, Hkv):
    """afk_attn_decode_fused: ONE launch - the last chunk block of every (batch, head) pair merges the chunks - must give exactly the two-launch
    result (same fold, chunk order), leave its arrival counters at zero, and do so on repeated calls into the same workspace (HIP-graph replay)"""
    from audio_flamingo_amd import _lib
    ops = _ops()
    B, Smax = 2, 1300
    spad = ops.pad64(Smax)
    nk, nq = Hkv * D, Hq * D
    q = _rand((B, nq), dev, 1.0, 1).to(BF)
    kc = _rand((B, Smax, nk), dev, 1.0, 2).to(BF)
    vt = _rand((B, Hkv, D, spad), dev, 1.0, 3).to(BF)
    kr = torch.tensor([[0, 1300], [41, 900]], device=dev, dtype=torch.int32)
    for ns in (8, 3, 1):
        nws = _lib.load().afk_attn_decode_workspace_floats(B, Hq, D, ns)
        o2 = torch.empty((B, nq), device=dev, dtype=BF)
        ws2 = torch.empty(nws, device=dev, dtype=torch.float32)
        args = lambda o, ws: (q.data_ptr(), nq, D, kc.data_ptr(), Smax * nk, nk, D, vt.data_ptr(), Hkv * D * spad, spad, o.data_ptr(), nq, D,
                              kr.data_ptr(), B, Hq, Hkv, D, float(D ** -0.5), ns, ws.data_ptr(), ops._stream())
        _lib.call("afk_attn_decode", *args(o2, ws2))
        ws1 = torch.zeros(nws, device=dev, dtype=torch.float32)
        for rep in range(4):
            o1 = torch.full((B, nq), 7.0, device=dev, dtype=BF)
            _lib.call("afk_attn_decode_fused", *args(o1, ws1))
            torch.cuda.synchronize()
            assert torch.equal(o1, o2), (ns, rep, float((o1.float() - o2.float()).abs().max()))
            assert int(ws1[-B * Hq:].view(torch.int32).abs().sum()) == 0, "arrival counters must be left at zero"


@pytest.mark.parametrize("B,S,Hq,Hkv,D,causal", [(8, 1024, 28, 4, 128, True), (2, 1536, 20, 20, 64, False), (1, 128, 4, 2, 128, True), (3, 384, 4, 4, 64, True)])
def test_attention_forward_persistent_form_bit_equal(dev, B, S, Hq, Hkv, D, causal):
    """round 5, opt-in (AFK_ATTN_PERSIST=1): the forward on resident blocks that pull (sample, head, query block) items from an atomic queue and overlap the next
    item's cold loads with the current item's last tile and store tail (attn_fwd_persist_kernel) - O and LSE bit-identical to the grid form, the queue words
    left at zero, repeated launches; with fewer items than resident blocks (every block takes exactly one) and with more"""
    ops = _ops()
    qkv = _rand((B * S, (Hq + 2 * Hkv) * D), dev, 0.5, 5).to(BF)
    old = ops.ATTN_PERSIST
    try:
        ops.ATTN_PERSIST = False
        o0, l0 = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
        ops.ATTN_PERSIST = True
        for _ in range(3):
            o1, l1 = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
            torch.cuda.synchronize()
            assert torch.equal(o0, o1) and torch.equal(l0[..., :S], l1[..., :S])
            assert ops._attn_queue(dev).tolist() == [0, 0]
        # round 6: the same kernel without its queue - block k runs items k and total - 1 - k (the paired-tile causal schedule, afk_attn_set_persist_paired)
        from audio_flamingo_amd import _lib

        _lib.call("afk_attn_set_persist_paired", 1)
        try:
            for _ in range(2):
                o2, l2 = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
                torch.cuda.synchronize()
                assert torch.equal(o0, o2) and torch.equal(l0[..., :S], l2[..., :S])
                assert ops._attn_queue(dev).tolist() == [0, 0]
        finally:
            _lib.call("afk_attn_set_persist_paired", 0)
    finally:
        ops.ATTN_PERSIST = old


@pytest.mark.parametrize("B,S,Hq,Hkv,D,causal,pad", [(8, 1024, 28, 4, 128, True, None), (2, 1500, 20, 20, 64, False, None), (2, 1000, 28, 4, 128, True, "right"),
                                                     (3, 777, 8, 8, 64, False, "right"), (2, 640, 8, 2, 128, True, "left"), (1, 128, 4, 2, 128, True, None),
                                                     (1, 2113, 28, 4, 128, True, "left"), (1, 96, 4, 4, 64, False, None), (3, 520, 6, 3, 64, True, None),
                                                     (1, 7774, 28, 4, 128, True, None), (5, 300, 7, 7, 128, False, "right")])
def test_attention_schedules_bit_equal(dev, B, S, Hq, Hkv, D, causal, pad):
    """round 6: the forward and dQ kernels' explicit-ring schedule (afk_attn_set_sched(1): row fragments through counted rings of opaque ds_read_b128 groups, the
    dQ tile as two 32-key halves, LDS-DMA pieces between the MFMA groups) computes the SAME MFMAs on the same operands in the same order as the
    compiler-scheduled form of rounds 1-5 (0): O, LSE and every gradient bit-identical - interior, diagonal, ragged, right- and left-padded tiles."""
    from audio_flamingo_amd import _lib

    ops = _ops()
    qkv = _rand((B * S, (Hq + 2 * Hkv) * D), dev, 0.5, 11).to(BF)
    do = _rand((B * S, Hq * D), dev, 0.5, 12).to(BF)
    kv_len = kv_lo = None
    if pad == "right":
        kv_len = torch.tensor([S - 37 * (i + 1) for i in range(B)], device=dev, dtype=torch.int32)
    elif pad == "left":
        kv_lo = torch.tensor([29 + 70 * i for i in range(B)], device=dev, dtype=torch.int32)
    res = {}
    try:
        # (schedule, XCD-aware block map): the map (afk_attn_set_xcd_map) only changes WHICH block computes a (sample, head, query block) - same bits again
        for key in ((0, 0), (1, 0), (1, 1), (0, 1), (1, 1)):
            _lib.call("afk_attn_set_sched", key[0])
            _lib.call("afk_attn_set_xcd_map", key[1])
            o, lse = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal, kv_len=kv_len, kv_lo=kv_lo)
            dqkv = ops.attn_bwd(qkv, o, do, lse, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal, kv_len=kv_len, kv_lo=kv_lo)
            torch.cuda.synchronize()
            if key in res:
                assert torch.equal(res[key][0], o) and torch.equal(res[key][2], dqkv)   # and deterministic run to run
            res[key] = (o.clone(), lse[..., :S].clone(), dqkv.clone())
    finally:
        _lib.call("afk_attn_set_sched", 1)
        _lib.call("afk_attn_set_xcd_map", 1)
    base = res[(0, 0)]
    assert torch.isfinite(base[0].float()).all() and torch.isfinite(base[2].float()).all()
    for key, r in res.items():
        assert torch.equal(base[0], r[0]), f"O differs: (schedule, xcd map) = {key}"
        assert torch.equal(base[1], r[1]), f"LSE differs: (schedule, xcd map) = {key}"
        assert torch.equal(base[2], r[2]), f"dQKV differs: (schedule, xcd map) = {key}"


@pytest.mark.parametrize("B,S,Hq,Hkv,D,parts,with_pos,pad", [(8, 1024, 28, 4, 128, 0, False, None), (2, 1000, 28, 4, 128, 7, True, "left"), (2, 384, 8, 8, 128, 0, True, None),
                                                              (3, 520, 8, 2, 64, 1, False, "right"), (1, 2113, 28, 4, 128, 3, True, None), (2, 200, 4, 1, 128, 2, False, None)])
def test_attention_backward_fused_rope_bit_equal(dev, B, S, Hq, Hkv, D, parts, with_pos, pad):
    """round 6: the rotary backward inside the attention backward (afk_attn2_bwd_fused_rope: dQ rotated in the dQ kernel's epilogue, dK in the combined GQA
    reduce - or in the sweep's epilogue when it writes the final dK: MHA, one part) gives EXACTLY the bits of afk_attn2_bwd_fused followed by
    afk_rope_inplace(backward) on the q | k columns; dV untouched.  GQA with 7 / 3 / 2 / 1 parts, MHA, explicit positions (left-padded rows), head_dim 64."""
    from audio_flamingo_amd import _lib

    ops = _ops()
    qkv = _rand((B * S, (Hq + 2 * Hkv) * D), dev, 0.5, 21).to(BF)
    do = _rand((B * S, Hq * D), dev, 0.5, 22).to(BF)
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, device=dev, dtype=torch.float32) / D))
    fr = torch.arange(S + 8, device=dev, dtype=torch.float32)[:, None] * inv[None]
    emb = torch.cat([fr, fr], -1)
    cos, sin = emb.cos().to(BF).contiguous(), emb.sin().to(BF).contiguous()
    kv_len = kv_lo = pos = None
    if pad == "right":
        kv_len = torch.tensor([S - 23 * (i + 1) for i in range(B)], device=dev, dtype=torch.int32)
    elif pad == "left":
        kv_lo = torch.tensor([17 + 50 * i for i in range(B)], device=dev, dtype=torch.int32)
    if with_pos:
        lo = kv_lo if kv_lo is not None else torch.zeros(B, device=dev, dtype=torch.int32)
        pos = (torch.arange(S, device=dev, dtype=torch.int32)[None, :] - lo[:, None]).clamp_min(0).reshape(-1).contiguous()
    o, lse = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=True, kv_len=kv_len, kv_lo=kv_lo)
    old = ops.ATTN_FUSE_ROPE_BWD
    try:
        _lib.call("afk_attn_set_dkdv_parts", parts)
        ops.ATTN_FUSE_ROPE_BWD = False
        ref = ops.attn_bwd(qkv, o, do, lse, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=True, kv_len=kv_len, kv_lo=kv_lo, rope=(cos, sin, pos))
        plain = ops.attn_bwd(qkv, o, do, lse, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=True, kv_len=kv_len, kv_lo=kv_lo)
        ops.ATTN_FUSE_ROPE_BWD = True
        for _ in range(2):
            got = ops.attn_bwd(qkv, o, do, lse, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=True, kv_len=kv_len, kv_lo=kv_lo, rope=(cos, sin, pos))
            torch.cuda.synchronize()
            assert torch.equal(got, ref), float((got.float() - ref.float()).abs().max())
    finally:
        ops.ATTN_FUSE_ROPE_BWD = old
        _lib.call("afk_attn_set_dkdv_parts", 0)
    nqk = (Hq + Hkv) * D
    assert torch.equal(got[:, nqk:], plain[:, nqk:]) and not torch.equal(got[:, :nqk], plain[:, :nqk])   # dV untouched, dq | dk really rotated


@pytest.mark.parametrize("M,S,Hq,Hkv,K,with_pos,with_bias", [(8192, 1024, 28, 4, 3584, False, True), (7774, 7774, 28, 4, 3584, True, True), (6000, 750, 14, 2, 1024, True, False),
                                                             (3000, 1000, 28, 4, 512, False, True)])
def test_gemm_rope_epilogue_bit_equal(dev, M, S, Hq, Hkv, K, with_pos, with_bias):
    """round 6: the qkv projection with the rotary embedding in the 256 x 256 GEMM's epilogue (afk_gemm_nt_bf16_rope: the two 64-column halves of a head sit in
    neighbouring waves, which trade their bf16-rounded Linear outputs through LDS) == afk_gemm_nt_bf16 + afk_rope_inplace bit for bit: q and k heads rotated,
    v columns untouched, ragged M, explicit positions, with and without bias."""
    ops = _ops()
    D = 128
    N, rc = (Hq + 2 * Hkv) * D, (Hq + Hkv) * D
    a = _rand((M, K), dev, 1.0, 31).to(BF)
    w = _rand((N, K), dev, K ** -0.5, 32).to(BF)
    bias = _rand((N,), dev, 0.5, 33).to(BF) if with_bias else None
    inv = 1.0 / (1e6 ** (torch.arange(0, D, 2, device=dev, dtype=torch.float32) / D))
    fr = torch.arange(S + 16, device=dev, dtype=torch.float32)[:, None] * inv[None]
    emb = torch.cat([fr, fr], -1)
    cos, sin = emb.cos().to(BF).contiguous(), emb.sin().to(BF).contiguous()
    pos = ((torch.arange(M, device=dev, dtype=torch.int32) * 7) % (S + 16)).contiguous() if with_pos else None
    old = ops.GEMM_FUSE_ROPE
    try:
        ops.GEMM_FUSE_ROPE = False
        ref = ops.gemm_nt_rope(a, w, bias, cos, sin, S=S, rope_cols=rc, D=D, pos=pos)
        plain = ops.gemm_nt(a, w, bias=bias)
        ops.GEMM_FUSE_ROPE = True
        for _ in range(2):
            got = ops.gemm_nt_rope(a, w, bias, cos, sin, S=S, rope_cols=rc, D=D, pos=pos)
            torch.cuda.synchronize()
            assert torch.equal(got, ref), float((got.float() - ref.float()).abs().max())
    finally:
        ops.GEMM_FUSE_ROPE = old
    assert torch.equal(got[:, rc:], plain[:, rc:]) and not torch.equal(got[:, :rc], plain[:, :rc])


@pytest.mark.parametrize("D,Hq,Hkv,B", [(128, 28, 4, 8), (64, 4, 2, 16), (128, 8, 2, 8)])
def test_attn_decode_group_kernel_bit_equal(dev, D, Hq, Hkv, B):
    """round 5: batched decode attention - ONE block per (sample, KV head, key chunk) serves all Hq / Hkv query heads (attn_decode_group_kernel, taken by
    afk_attn_decode_fused when B x Hkv x nsplit >= 128) - must be BIT-IDENTICAL to the per-head split + combine launches (afk_attn_decode), with ragged and
    left-padded key ranges, an empty chunk, repeated calls into the same workspace and counters left at zero"""
    from audio_flamingo_amd import _lib
    ops = _ops()
    Smax = 1300
    spad = ops.pad64(Smax)
    nk, nq = Hkv * D, Hq * D
    q = _rand((B, nq), dev, 1.0, 1).to(BF)
    kc = _rand((B, Smax, nk), dev, 1.0, 2).to(BF)
    vt = _rand((B, Hkv, D, spad), dev, 1.0, 3).to(BF)
    kr = torch.tensor([[(13 * b) % 50, max(1300 - 150 * b, 60)] for b in range(B)], device=dev, dtype=torch.int32)
    kr[-1] = torch.tensor([5, 9], device=dev, dtype=torch.int32)     # four visible keys: most chunks of this sample are empty
    _lib.call("afk_attn_decode_set_group", 1)   # the group form is opt-in (measured 1-2 % slower on the B = 8 step): select it for this test
    try:
        _group_checks(_lib, ops, dev, q, kc, vt, kr, B, Hq, Hkv, D, Smax, spad, nk, nq)
    finally:
        _lib.call("afk_attn_decode_set_group", -1)


@pytest.mark.parametrize("D,Hq,Hkv,B", [(128, 28, 4, 8), (128, 28, 4, 2), (64, 16, 4, 8), (128, 8, 4, 5), (64, 8, 1, 3)])
def test_attn_decode_matrix_pipe_group_form(dev, D, Hq, Hkv, B):
    """round 6: the group form on the matrix pipe (attn_decode_gmma_kernel, afk_attn_decode_set_group(3)): K.Q^T and Vt.P as v_mfma_f32_32x32x16_bf16 chains, probabilities
    rounded to bf16 as the product reads them.  Against the per-head launches (same inputs: equal within bf16 rounding of the probabilities) and against fp32 softmax
    attention; ragged / left-padded ranges, a nearly empty sample, a key range that ends at the last cache slot, repeated calls, counters left at zero, determinism"""
    from audio_flamingo_amd import _lib
    ops = _ops()
    Smax = 1280   # a multiple of 64: spad == Smax, the last chunk ends at the row's end
    spad = ops.pad64(Smax)
    nk, nq = Hkv * D, Hq * D
    q = _rand((B, nq), dev, 1.0, 1).to(BF)
    kc = _rand((B, Smax, nk), dev, 1.0, 2).to(BF)
    vt = _rand((B, Hkv, D, spad), dev, 1.0, 3).to(BF)
    vt[:, :, :, 1275:] = float("nan")   # behind the last key of sample 1: must never reach a sum
    kr = torch.tensor([[(13 * b) % 50, max(1280 - 150 * b, 60)] for b in range(B)], device=dev, dtype=torch.int32)
    kr[0] = torch.tensor([0, 1280], device=dev, dtype=torch.int32)
    vt[0] = _rand((Hkv, D, spad), dev, 1.0, 4).to(BF)
    if B > 1:
        kr[1] = torch.tensor([3, 1275], device=dev, dtype=torch.int32)
    kr[-1] = torch.tensor([5, 9], device=dev, dtype=torch.int32)
    for ns in (8, 4):
        nws = _lib.load().afk_attn_decode_workspace_floats(B, Hq, D, ns)
        args = lambda o, ws: (q.data_ptr(), nq, D, kc.data_ptr(), Smax * nk, nk, D, vt.data_ptr(), Hkv * D * spad, spad, o.data_ptr(), nq, D,
                              kr.data_ptr(), B, Hq, Hkv, D, float(D ** -0.5), ns, ws.data_ptr(), ops._stream())
        o2 = torch.empty((B, nq), device=dev, dtype=BF)
        _lib.call("afk_attn_decode", *args(o2, torch.empty(nws, device=dev, dtype=torch.float32)))
        ws1 = torch.zeros(nws, device=dev, dtype=torch.float32)
        outs = []
        _lib.call("afk_attn_decode_set_group", 3)
        try:
            for rep in range(3):
                o1 = torch.full((B, nq), 7.0, device=dev, dtype=BF)
                _lib.call("afk_attn_decode_fused", *args(o1, ws1))
                torch.cuda.synchronize()
                outs.append(o1)
                assert int(ws1[-B * Hq:].view(torch.int32).abs().sum()) == 0, "arrival counters must be left at zero"
        finally:
            _lib.call("afk_attn_decode_set_group", -1)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        assert bool(torch.isfinite(outs[0].float()).all())
        _cmp(f"matrix-pipe group form vs per-head ns={ns}", outs[0], o2.float(), atol=2e-2, rtol=2e-2)
    for b in range(B):
        lo, hi = int(kr[b, 0]), int(kr[b, 1])
        qq = q[b].float().view(Hq, D)
        kk = kc[b, lo:hi].float().view(hi - lo, Hkv, D).repeat_interleave(Hq // Hkv, 1)
        vv = vt[b, :, :, lo:hi].float().permute(2, 0, 1).repeat_interleave(Hq // Hkv, 1)
        pr = torch.softmax(torch.einsum("hd,shd->hs", qq, kk) * D ** -0.5, -1)
        _cmp(f"matrix-pipe group decode attention b={b}", outs[0][b], torch.einsum("hs,shd->hd", pr, vv).reshape(-1), atol=2e-2, rtol=2e-2)


def _group_checks(_lib, ops, dev, q, kc, vt, kr, B, Hq, Hkv, D, Smax, spad, nk, nq):
    for ns in (8, 4) if B * Hkv * 4 >= 128 else (8,):
        nws = _lib.load().afk_attn_decode_workspace_floats(B, Hq, D, ns)
        args = lambda o, ws: (q.data_ptr(), nq, D, kc.data_ptr(), Smax * nk, nk, D, vt.data_ptr(), Hkv * D * spad, spad, o.data_ptr(), nq, D,
                              kr.data_ptr(), B, Hq, Hkv, D, float(D ** -0.5), ns, ws.data_ptr(), ops._stream())
        o2 = torch.empty((B, nq), device=dev, dtype=BF)
        _lib.call("afk_attn_decode", *args(o2, torch.empty(nws, device=dev, dtype=torch.float32)))
        ws1 = torch.zeros(nws, device=dev, dtype=torch.float32)
        for rep in range(3):
            o1 = torch.full((B, nq), 7.0, device=dev, dtype=BF)
            _lib.call("afk_attn_decode_fused", *args(o1, ws1))
            torch.cuda.synchronize()
            assert torch.equal(o1, o2), (ns, rep, float((o1.float() - o2.float()).abs().max()))
            assert int(ws1[-B * Hq:].view(torch.int32).abs().sum()) == 0, "arrival counters must be left at zero"
    # and against plain fp32 softmax attention on the visible interval
    for b in (0, B - 1):
        lo, hi = int(kr[b, 0]), int(kr[b, 1])
        qq = q[b].float().view(Hq, D)
        kk = kc[b, lo:hi].float().view(hi - lo, Hkv, D).repeat_interleave(Hq // Hkv, 1)
        vv = vt[b, :, :, lo:hi].float().permute(2, 0, 1).repeat_interleave(Hq // Hkv, 1)
        pr = torch.softmax(torch.einsum("hd,shd->hs", qq, kk) * D ** -0.5, -1)
        _cmp(f"group decode attention b={b}", o2[b], torch.einsum("hs,shd->hd", pr, vv).reshape(-1), atol=2e-2, rtol=2e-2)


def test_last_block_hand_over_under_memory_load(dev):
    """Stress of the lock-free "last block merges" hand-over (afk_colsum_bf16_fused, afk_attn_decode_fused): the partials are agent-scope write-through
    stores that every storing wave drains with an explicit s_waitcnt vmcnt(0) before the block barrier and the counter bump (the workgroup-scope
    fence that used to stand there emits no wait on gfx950).  Hundreds of launches on two streams while a third stream saturates HBM with copies, fresh
    (poisoned) workspaces every launch so that a partial read before it landed shows up as a wrong - not merely stale-but-equal - result."""
    from audio_flamingo_amd import _lib
    ops = _ops()
    load_stream, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    big_a = torch.empty(1 << 28, device=dev, dtype=torch.float32)   # 1 GiB each way: far beyond the 8 x 4 MiB of L2 and the 256 MiB of MALL
    big_b = torch.empty_like(big_a)
    rows, cols = 8192, 4608
    x = _rand((rows, cols), dev, seed=11).to(BF)
    try:
        ops.COLSUM_FUSED = False
        want = ops.colsum(x, torch.zeros(cols, device=dev, dtype=BF)).clone()
    finally:
        ops.COLSUM_FUSED = True
    ns = _lib.load().afk_colsum_slices(rows)
    cnt = {None: torch.zeros(4096, device=dev, dtype=torch.int32), s2: torch.zeros(4096, device=dev, dtype=torch.int32)}
    # decode attention at the 7B head geometry
    B, Smax, D, Hq, Hkv, nsplit = 8, 1300, 128, 28, 4, 8
    spad = ops.pad64(Smax)
    nk, nq = Hkv * D, Hq * D
    q = _rand((B, nq), dev, 1.0, 1).to(BF)
    kc = _rand((B, Smax, nk), dev, 1.0, 2).to(BF)
    vt = _rand((B, Hkv, D, spad), dev, 1.0, 3).to(BF)
    kr = torch.tensor([[0, 1300 - 37 * b] for b in range(B)], device=dev, dtype=torch.int32)
    nws = _lib.load().afk_attn_decode_workspace_floats(B, Hq, D, nsplit)
    dargs = lambda o, ws, st: (q.data_ptr(), nq, D, kc.data_ptr(), Smax * nk, nk, D, vt.data_ptr(), Hkv * D * spad, spad, o.data_ptr(), nq, D,
                               kr.data_ptr(), B, Hq, Hkv, D, float(D ** -0.5), nsplit, ws.data_ptr(), st)
    o_want = torch.empty((B, nq), device=dev, dtype=BF)
    _lib.call("afk_attn_decode", *dargs(o_want, torch.empty(nws, device=dev, dtype=torch.float32), ops._stream()))
    torch.cuda.synchronize()
    _lib.call("afk_attn_decode_set_group", 0)   # the per-head form: bit-comparable with the two-launch result (a launch this size takes the matrix-pipe group form by default)
    N = 150
    outs, douts = [], []
    with torch.cuda.stream(load_stream):
        for _ in range(60):
            big_b.copy_(big_a)
            big_a.copy_(big_b)
    for i in range(N):
        for st in (None, s2):
            ctx = torch.cuda.stream(st) if st is not None else torch.cuda.stream(torch.cuda.current_stream())
            with ctx:
                ws = torch.full((ns * cols,), float("nan"), device=dev, dtype=torch.float32)
                out = torch.zeros(cols, device=dev, dtype=BF)
                _lib.call("afk_colsum_bf16_fused", x.data_ptr(), x.stride(0), rows, cols, out.data_ptr(), 0, ws.data_ptr(), cnt[st].data_ptr(), ops._stream())
                outs.append(out)
                dws = torch.full((nws,), float("nan"), device=dev, dtype=torch.float32)
                dws[-B * Hq:].view(torch.int32).zero_()
                o = torch.full((B, nq), 7.0, device=dev, dtype=BF)
                _lib.call("afk_attn_decode_fused", *dargs(o, dws, ops._stream()))
                douts.append((o, dws))
    torch.cuda.synchronize()
    bad = sum(int(not torch.equal(o, want)) for o in outs)
    dbad = sum(int(not torch.equal(o, o_want)) for o, _ in douts)
    assert bad == 0 and dbad == 0, f"hand-over lost partials: colsum {bad}/{len(outs)}, decode attention {dbad}/{len(douts)} launches differ"
    _lib.call("afk_attn_decode_set_group", -1)
    assert all(int(w[-B * Hq:].view(torch.int32).abs().sum()) == 0 for _, w in douts) and all(int(c.abs().sum()) == 0 for c in cnt.values())


@pytest.mark.parametrize("knobs", ["", "S=1,1,1,1,1", "S=2,2,2,2,2 R=4,4,4,4,4", "S=8,8,8,8,8", "S=4,4,4,4,4 R=4,4,4,4,4"])
def test_decode_chain_kernels_vs_standalone_sequence(dev, knobs, monkeypatch):
    """csrc/decode_chain.hip (one launch per Linear of a single-sequence decode step, RMSNorm in the consumer's prologue, bias / RoPE / cache append /
    residual / SwiGLU in the producer's epilogue) against the stand-alone kernel sequence with the same rounding points, at the AF3-7B widths"""
    from audio_flamingo_amd import _lib
    ops = _ops()
    for kv in knobs.split():   # rows per group / waves per group of every launch form (csrc/decode_chain.hip chain_knob)
        monkeypatch.setenv("AFK_CHAIN_" + kv[0], kv[2:])
    H, Hq, Hkv, D, I = 3584, 28, 4, 128, 18944
    nq, nk = Hq * D, Hkv * D
    N = nq + 2 * nk
    st = ops._stream()
    x = _rand((1, H), dev, 1.0, 1).to(BF)
    nw = (1 + 0.1 * _rand((H,), dev, seed=2)).to(BF)
    w = _rand((N, H), dev, 0.02, 3).to(BF)
    bias = _rand((N,), dev, 0.1, 4).to(BF)
    Smax, pos, start = 256, 37, 41
    spad = ops.pad64(Smax)
    inv = 1.0 / (1e6 ** (torch.arange(0, D, 2, device=dev, dtype=torch.float32) / D))
    fr = torch.arange(64, device=dev, dtype=torch.float32)[:, None] * inv[None]
    emb = torch.cat([fr, fr], -1)
    cos, sin = emb.cos().to(BF).contiguous(), emb.sin().to(BF).contiguous()
    pos_t = torch.tensor([pos], device=dev, dtype=torch.int32)
    start_t = torch.tensor([start], device=dev, dtype=torch.int32)
    # reference: rmsnorm -> gemm + bias -> rope -> cache append, on the stand-alone kernels
    h, _ = ops.rmsnorm_fwd(x, nw, 1e-6)
    qkv = ops.gemm_nt(h, w, bias=bias)
    ops.rope_(qkv, cos, sin, S=1, nheads=Hq + Hkv, D=D, pos=pos_t)
    Kc = torch.zeros((1, Smax, nk), device=dev, dtype=BF)
    Vt = torch.zeros((1, Hkv, D, spad), device=dev, dtype=BF)
    q = torch.zeros((1, nq), device=dev, dtype=BF)
    _lib.call("afk_decode_chain_qkv", x.data_ptr(), nw.data_ptr(), 1e-6, w.data_ptr(), w.stride(0), H, bias.data_ptr(), cos.data_ptr(), sin.data_ptr(),
              pos_t.data_ptr(), q.data_ptr(), Kc.data_ptr(), Vt.data_ptr(), spad, start_t.data_ptr(), Hq, Hkv, D, st)
    torch.cuda.synchronize()
    _cmp("chain q", q[0], qkv[0, :nq].float(), atol=3e-2, rtol=2e-2)
    _cmp("chain k", Kc[0, start], qkv[0, nq:nq + nk].float(), atol=3e-2, rtol=2e-2)
    _cmp("chain v", Vt[0, :, :, start].reshape(-1), qkv[0, nq + nk:].float(), atol=3e-2, rtol=2e-2)
    assert float(Kc.float().abs().sum() - Kc[0, start].float().abs().sum()) == 0.0, "only the new cache slot may be written"
    # o_proj / down: linear + residual
    for K_ in (nq, I):
        a = _rand((1, K_), dev, 1.0, 5).to(BF)
        wl = _rand((H, K_), dev, 0.02, 6).to(BF)
        out = torch.empty((1, H), device=dev, dtype=BF)
        _lib.call("afk_decode_chain_linear_residual", a.data_ptr(), wl.data_ptr(), wl.stride(0), H, K_, x.data_ptr(), out.data_ptr(), st)
        _cmp(f"chain linear+residual K={K_}", out, ops.gemm_nt(a, wl, residual=x).float(), atol=3e-2, rtol=2e-2)
    # gate|up: rmsnorm prologue + SwiGLU epilogue
    wgu = _rand((2 * I, H), dev, 0.02, 7).to(BF)
    act = torch.empty((1, I), device=dev, dtype=BF)
    _lib.call("afk_decode_chain_gate_up", x.data_ptr(), nw.data_ptr(), 1e-6, wgu.data_ptr(), wgu.stride(0), I, H, act.data_ptr(), st)
    _cmp("chain gate|up", act, ops.silu_mul_fwd(ops.gemm_nt(h, wgu)).float(), atol=3e-2, rtol=3e-2)
    # final norm + lm_head: fp32 logits holding bf16 values
    V = 4096
    wh = _rand((V, H), dev, 0.02, 8).to(BF)
    logits = torch.empty((1, V), device=dev, dtype=torch.float32)
    pv = torch.empty(V // 8, device=dev, dtype=torch.float32)
    pi = torch.empty(V // 8, device=dev, dtype=torch.int32)
    _lib.call("afk_decode_chain_lm_head", x.data_ptr(), nw.data_ptr(), 1e-6, wh.data_ptr(), wh.stride(0), V, H, logits.data_ptr(), pv.data_ptr(), pi.data_ptr(), st)
    assert torch.equal(logits, logits.to(BF).float())
    _cmp("chain lm_head", logits, ops.gemm_nt(h, wh).float(), atol=3e-2, rtol=2e-2)
    # greedy selection on the device: per-group (max, argmax) + the select launch == torch.argmax (ties: lowest index), state advanced, embedding row fetched
    assert torch.equal(pv, logits.view(V // 8, 8).max(-1).values)
    assert torch.equal(pi.long(), logits.view(V // 8, 8).argmax(-1) + 8 * torch.arange(V // 8, device=dev))
    emb = _rand((V, H), dev, 1.0, 9).to(BF)
    for tie in (False, True):
        if tie:   # the maximum twice: the lower row wins
            top = int(logits.argmax())
            wh[(top + 1234) % V] = wh[top]
            _lib.call("afk_decode_chain_lm_head", x.data_ptr(), nw.data_ptr(), 1e-6, wh.data_ptr(), wh.stride(0), V, H, logits.data_ptr(), pv.data_ptr(), pi.data_ptr(), st)
            assert int((logits == logits.max()).sum()) >= 2
        nxt = torch.full((1,), -1, device=dev, dtype=torch.int64)
        toks = torch.full((8,), -1, device=dev, dtype=torch.int64)
        state = torch.tensor([3, 41, 40, 37], device=dev, dtype=torch.int32)
        x0 = torch.zeros((1, H), device=dev, dtype=BF)
        _lib.call("afk_decode_select_greedy", pv.data_ptr(), pi.data_ptr(), V // 8, nxt.data_ptr(), toks.data_ptr(), 5 - 40, state.data_ptr(), emb.data_ptr(), emb.stride(0), H,
                  x0.data_ptr(), st)
        want = int((logits[0] == logits[0].max()).nonzero()[0])
        assert int(nxt) == want and toks.tolist() == [-1] * 5 + [want, -1, -1] and state.tolist() == [3, 42, 41, 38]
        assert torch.equal(x0[0], emb[want])


@pytest.mark.parametrize("mfma", ["1", "0"])
@pytest.mark.parametrize("M", [2, 3, 8])
def test_decode_chain_batched_kernels_vs_standalone_sequence(dev, M, mfma, monkeypatch):
    """the decode-chain launches for M = 2 .. 8 sequences per step (M input rows against every weight vector, inputs already normalised) against the
    stand-alone kernel sequence at the AF3-7B widths, in both forms: 32-row groups on the matrix pipe (the weights are the MFMA's A operand straight from
    global memory) and 8-row groups on v_dot2 (AFK_CHAIN_MFMA=0; also what odd shapes take).  Rows of different sequences do not mix."""
    from audio_flamingo_amd import _lib
    ops = _ops()
    monkeypatch.setenv("AFK_CHAIN_MFMA", mfma)
    H, Hq, Hkv, D, I = 3584, 28, 4, 128, 18944
    nq, nk = Hq * D, Hkv * D
    N = nq + 2 * nk
    st = ops._stream()
    h = _rand((M, H), dev, 1.0, 1).to(BF)
    res = _rand((M, H), dev, 1.0, 2).to(BF)
    w = _rand((N, H), dev, 0.02, 3).to(BF)
    bias = _rand((N,), dev, 0.1, 4).to(BF)
    Smax, start = 256, 41
    spad = ops.pad64(Smax)
    inv = 1.0 / (1e6 ** (torch.arange(0, D, 2, device=dev, dtype=torch.float32) / D))
    fr = torch.arange(64, device=dev, dtype=torch.float32)[:, None] * inv[None]
    emb = torch.cat([fr, fr], -1)
    cos, sin = emb.cos().to(BF).contiguous(), emb.sin().to(BF).contiguous()
    pos_t = torch.tensor([37 - 3 * m for m in range(M)], device=dev, dtype=torch.int32)   # left-padded rows sit at different positions
    start_t = torch.tensor([start], device=dev, dtype=torch.int32)
    qkv = ops.gemm_nt(h, w, bias=bias)
    ops.rope_(qkv, cos, sin, S=1, nheads=Hq + Hkv, D=D, pos=pos_t)
    Kc = torch.zeros((M, Smax, nk), device=dev, dtype=BF)
    Vt = torch.zeros((M, Hkv, D, spad), device=dev, dtype=BF)
    q = torch.zeros((M, nq), device=dev, dtype=BF)
    _lib.call("afk_decode_chain_qkv_batched", h.data_ptr(), h.stride(0), M, w.data_ptr(), w.stride(0), H, bias.data_ptr(), cos.data_ptr(), sin.data_ptr(), pos_t.data_ptr(),
              q.data_ptr(), nq, Kc.data_ptr(), Smax * nk, Vt.data_ptr(), Hkv * D * spad, spad, start_t.data_ptr(), Hq, Hkv, D, st)
    torch.cuda.synchronize()
    _cmp("batched q", q, qkv[:, :nq].float(), atol=3e-2, rtol=2e-2)
    _cmp("batched k", Kc[:, start], qkv[:, nq:nq + nk].float(), atol=3e-2, rtol=2e-2)
    _cmp("batched v", Vt[:, :, :, start].reshape(M, -1), qkv[:, nq + nk:].float(), atol=3e-2, rtol=2e-2)
    Kc[:, start] = 0
    Vt[:, :, :, start] = 0
    assert float(Kc.float().abs().sum()) == 0.0 and float(Vt.float().abs().sum()) == 0.0, "only the new cache slot may be written"
    for K_ in (nq, I):
        a = _rand((M, K_), dev, 1.0, 5).to(BF)
        wl = _rand((H, K_), dev, 0.02, 6).to(BF)
        out = torch.empty((M, H), device=dev, dtype=BF)
        _lib.call("afk_decode_chain_linear_residual_batched", a.data_ptr(), K_, M, wl.data_ptr(), wl.stride(0), H, K_, res.data_ptr(), H, out.data_ptr(), H, st)
        _cmp(f"batched linear+residual K={K_}", out, ops.gemm_nt(a, wl, residual=res).float(), atol=3e-2, rtol=2e-2)
        solo = torch.empty((1, H), device=dev, dtype=BF)   # the same launch on the last row alone: rows of different sequences do not mix
        _lib.call("afk_decode_chain_linear_residual_batched", a[M - 1:].data_ptr(), K_, 1, wl.data_ptr(), wl.stride(0), H, K_, res[M - 1:].data_ptr(), H, solo.data_ptr(), H, st)
        assert torch.equal(solo[0], out[M - 1])
    wgu = _rand((2 * I, H), dev, 0.02, 7).to(BF)
    act = torch.empty((M, I), device=dev, dtype=BF)
    _lib.call("afk_decode_chain_gate_up_batched", h.data_ptr(), h.stride(0), M, wgu.data_ptr(), wgu.stride(0), I, H, act.data_ptr(), I, st)
    _cmp("batched gate|up", act, ops.silu_mul_fwd(ops.gemm_nt(h, wgu)).float(), atol=3e-2, rtol=3e-2)
    V = 4096
    wh = _rand((V, H), dev, 0.02, 8).to(BF)
    logits = torch.empty((M, V), device=dev, dtype=torch.float32)
    _lib.call("afk_decode_chain_lm_head_batched", h.data_ptr(), h.stride(0), M, wh.data_ptr(), wh.stride(0), V, H, logits.data_ptr(), V, st)
    assert torch.equal(logits, logits.to(BF).float())
    _cmp("batched lm_head", logits, ops.gemm_nt(h, wh).float(), atol=3e-2, rtol=2e-2)


@pytest.mark.parametrize("M", [1, 4, 8, 9, 13, 16, 19, 24, 32])
def test_decode_chain_norm_in_prologue_batched(dev, M):
    """afk_decode_chain_{qkv,gate_up,lm_head}_norm_batched (RMSNorm taken in the Linear's own prologue, rows normalised through a wave-private LDS strip) against
    afk_rmsnorm_fwd + the plain matrix-pipe launches at the AF3-7B widths: equal up to the fp32 summation order of the row statistic (a differing last bit of a
    normalised bf16 value here and there), rows of different sequences do not mix, only the new cache slot is written"""
    from audio_flamingo_amd import _lib
    ops = _ops()
    H, Hq, Hkv, D, I = 3584, 28, 4, 128, 18944
    nq, nk = Hq * D, Hkv * D
    N = nq + 2 * nk
    st = ops._stream()
    x = _rand((M, H), dev, 1.0, 1).to(BF)
    nw = (1 + 0.1 * _rand((H,), dev, seed=2)).to(BF)
    h, _ = ops.rmsnorm_fwd(x, nw, 1e-6)
    w = _rand((N, H), dev, 0.02, 3).to(BF)
    bias = _rand((N,), dev, 0.1, 4).to(BF)
    Smax, start = 256, 41
    spad = ops.pad64(Smax)
    inv = 1.0 / (1e6 ** (torch.arange(0, D, 2, device=dev, dtype=torch.float32) / D))
    fr = torch.arange(64, device=dev, dtype=torch.float32)[:, None] * inv[None]
    emb = torch.cat([fr, fr], -1)
    cos, sin = emb.cos().to(BF).contiguous(), emb.sin().to(BF).contiguous()
    pos_t = torch.tensor([37 - 3 * (m % 12) for m in range(M)], device=dev, dtype=torch.int32)
    start_t = torch.tensor([start], device=dev, dtype=torch.int32)
    qkv = ops.gemm_nt(h, w, bias=bias)
    ops.rope_(qkv, cos, sin, S=1, nheads=Hq + Hkv, D=D, pos=pos_t)
    Kc = torch.zeros((M, Smax, nk), device=dev, dtype=BF)
    Vt = torch.zeros((M, Hkv, D, spad), device=dev, dtype=BF)
    q = torch.zeros((M, nq), device=dev, dtype=BF)
    _lib.call("afk_decode_chain_qkv_norm_batched", x.data_ptr(), H, M, nw.data_ptr(), 1e-6, w.data_ptr(), w.stride(0), H, bias.data_ptr(), cos.data_ptr(), sin.data_ptr(),
              pos_t.data_ptr(), q.data_ptr(), nq, Kc.data_ptr(), Smax * nk, Vt.data_ptr(), Hkv * D * spad, spad, start_t.data_ptr(), Hq, Hkv, D, None, 0, st)
    torch.cuda.synchronize()
    _cmp("norm-prologue q", q, qkv[:, :nq].float(), atol=3e-2, rtol=2e-2)
    _cmp("norm-prologue k", Kc[:, start], qkv[:, nq:nq + nk].float(), atol=3e-2, rtol=2e-2)
    _cmp("norm-prologue v", Vt[:, :, :, start].reshape(M, -1), qkv[:, nq + nk:].float(), atol=3e-2, rtol=2e-2)
    Kc[:, start] = 0
    Vt[:, :, :, start] = 0
    assert float(Kc.float().abs().sum()) == 0.0 and float(Vt.float().abs().sum()) == 0.0, "only the new cache slot may be written"
    wgu = _rand((2 * I, H), dev, 0.02, 7).to(BF)
    act = torch.empty((M, I), device=dev, dtype=BF)
    _lib.call("afk_decode_chain_gate_up_norm_batched", x.data_ptr(), H, M, nw.data_ptr(), 1e-6, wgu.data_ptr(), wgu.stride(0), I, H, act.data_ptr(), I, None, 0, st)
    _cmp("norm-prologue gate|up", act, ops.silu_mul_fwd(ops.gemm_nt(h, wgu)).float(), atol=3e-2, rtol=3e-2)
    plain = torch.empty((M, I), device=dev, dtype=BF)   # the plain matrix-pipe launch on afk_rmsnorm_fwd's rows: the same products, so almost every value is bit-equal
    import os as _os
    _os.environ["AFK_CHAIN_MFMA"] = "1"
    try:
        for m0 in range(0, M, 8):   # the plain entry point takes eight sequences at a time
            mm = min(8, M - m0)
            _lib.call("afk_decode_chain_gate_up_batched", h[m0:].data_ptr(), H, mm, wgu.data_ptr(), wgu.stride(0), I, H, plain[m0:].data_ptr(), I, st)
    finally:
        del _os.environ["AFK_CHAIN_MFMA"]
    assert int((plain != act).sum()) <= M * I // 50, f"{int((plain != act).sum())} of {M * I} values differ from the norm-launch form"
    if M > 8:   # a group of eight is an independent instance of the eight-sequence launch: bit for bit
        for m0 in range(0, M, 8):
            mm = min(8, M - m0)
            part = torch.empty((mm, I), device=dev, dtype=BF)
            _lib.call("afk_decode_chain_gate_up_norm_batched", x[m0:].data_ptr(), H, mm, nw.data_ptr(), 1e-6, wgu.data_ptr(), wgu.stride(0), I, H, part.data_ptr(), I, None, 0, st)
            assert torch.equal(part, act[m0:m0 + mm]), f"group at sequence {m0} differs from the eight-sequence launch"
    solo = torch.empty((1, I), device=dev, dtype=BF)
    _lib.call("afk_decode_chain_gate_up_norm_batched", x[M - 1:].data_ptr(), H, 1, nw.data_ptr(), 1e-6, wgu.data_ptr(), wgu.stride(0), I, H, solo.data_ptr(), I, None, 0, st)
    assert torch.equal(solo[0], act[M - 1])
    # the statistic from the PRODUCER's partial sums: o_proj-like Linear + residual writes the rows and ss_part, the next Linear folds them
    a_in = _rand((M, nq), dev, 1.0, 11).to(BF)
    wl = _rand((H, nq), dev, 0.02, 12).to(BF)
    res = _rand((M, H), dev, 1.0, 13).to(BF)
    rows_w = torch.empty((M, H), device=dev, dtype=BF)
    for m0 in range(0, M, 8):
        mm = min(8, M - m0)
        _lib.call("afk_decode_chain_linear_residual_batched", a_in[m0:].data_ptr(), nq, mm, wl.data_ptr(), wl.stride(0), H, nq, res[m0:].data_ptr(), H, rows_w[m0:].data_ptr(), H, st)
    rows_s = torch.empty((M, H), device=dev, dtype=BF)
    NGt = 1 if M <= 8 else 2 if M <= 16 else 4   # groups the launch runs
    ssp = torch.full((NGt, 8, H // 16), float("nan"), device=dev, dtype=torch.float32)   # [groups][row][blocks of the launch]
    _lib.call("afk_decode_chain_linear_residual_ss_batched", a_in.data_ptr(), nq, M, wl.data_ptr(), wl.stride(0), H, nq, res.data_ptr(), H, rows_s.data_ptr(), H, ssp.data_ptr(), st)
    _cmp("linear+residual (ss form)", rows_s, rows_w.float(), atol=3e-2, rtol=2e-2)
    want_ss = (rows_s.float() ** 2).sum(1)
    got_all = ssp.sum(2).reshape(-1)               # [groups * 8]: sequence 8 g + m
    got_ss = got_all[:M]
    assert bool(((got_ss - want_ss).abs() <= 1e-4 * want_ss).all()) and float(got_all[M:].abs().sum()) == 0.0
    act_b = torch.empty((M, I), device=dev, dtype=BF)
    act_p = torch.empty((M, I), device=dev, dtype=BF)
    _lib.call("afk_decode_chain_gate_up_norm_batched", rows_s.data_ptr(), H, M, nw.data_ptr(), 1e-6, wgu.data_ptr(), wgu.stride(0), I, H, act_b.data_ptr(), I, None, 0, st)
    _lib.call("afk_decode_chain_gate_up_norm_batched", rows_s.data_ptr(), H, M, nw.data_ptr(), 1e-6, wgu.data_ptr(), wgu.stride(0), I, H, act_p.data_ptr(), I, ssp.data_ptr(), H // 16, st)
    assert int((act_b != act_p).sum()) <= M * I // 50, f"{int((act_b != act_p).sum())} of {M * I} values differ between the two sources of the statistic"
    _cmp("norm-prologue gate|up, producer's statistic", act_p, act_b.float(), atol=3e-2, rtol=3e-2)
    V = 4096
    wh = _rand((V, H), dev, 0.02, 8).to(BF)
    logits = torch.empty((M, V), device=dev, dtype=torch.float32)
    _lib.call("afk_decode_chain_lm_head_norm_batched", x.data_ptr(), H, M, nw.data_ptr(), 1e-6, wh.data_ptr(), wh.stride(0), V, H, logits.data_ptr(), V, None, 0, st)
    assert torch.equal(logits, logits.to(BF).float())
    _cmp("norm-prologue lm_head", logits, ops.gemm_nt(h, wh).float(), atol=3e-2, rtol=2e-2)


@pytest.mark.parametrize("M", [9, 16, 21, 32])
def test_decode_chain_batched_plain_entries_more_than_eight_sequences(dev, M):
    """the plain _batched entry points (inputs already normalised) for 9 .. 32 sequences: groups of eight as columns of one MFMA (gemv_chain_mfma_ng_kernel) - every
    group of eight rows bit-identical to... the stand-alone kernel sequence within tolerance, and to the eight-sequence launches of the same rows where those take the
    same arithmetic (the sums of a column do not depend on the other columns)"""
    from audio_flamingo_amd import _lib
    ops = _ops()
    H, Hq, Hkv, D, I = 3584, 28, 4, 128, 18944
    nq, nk = Hq * D, Hkv * D
    N = nq + 2 * nk
    st = ops._stream()
    h = _rand((M, H), dev, 1.0, 1).to(BF)
    res = _rand((M, H), dev, 1.0, 2).to(BF)
    w = _rand((N, H), dev, 0.02, 3).to(BF)
    bias = _rand((N,), dev, 0.1, 4).to(BF)
    Smax, start = 256, 41
    spad = ops.pad64(Smax)
    inv = 1.0 / (1e6 ** (torch.arange(0, D, 2, device=dev, dtype=torch.float32) / D))
    fr = torch.arange(64, device=dev, dtype=torch.float32)[:, None] * inv[None]
    emb = torch.cat([fr, fr], -1)
    cos, sin = emb.cos().to(BF).contiguous(), emb.sin().to(BF).contiguous()
    pos_t = torch.tensor([37 - 3 * (m % 12) for m in range(M)], device=dev, dtype=torch.int32)
    start_t = torch.tensor([start], device=dev, dtype=torch.int32)
    qkv = ops.gemm_nt(h, w, bias=bias)
    ops.rope_(qkv, cos, sin, S=1, nheads=Hq + Hkv, D=D, pos=pos_t)
    Kc = torch.zeros((M, Smax, nk), device=dev, dtype=BF)
    Vt = torch.zeros((M, Hkv, D, spad), device=dev, dtype=BF)
    q = torch.zeros((M, nq), device=dev, dtype=BF)
    _lib.call("afk_decode_chain_qkv_batched", h.data_ptr(), H, M, w.data_ptr(), w.stride(0), H, bias.data_ptr(), cos.data_ptr(), sin.data_ptr(), pos_t.data_ptr(),
              q.data_ptr(), nq, Kc.data_ptr(), Smax * nk, Vt.data_ptr(), Hkv * D * spad, spad, start_t.data_ptr(), Hq, Hkv, D, st)
    torch.cuda.synchronize()
    _cmp("grouped q", q, qkv[:, :nq].float(), atol=3e-2, rtol=2e-2)
    _cmp("grouped k", Kc[:, start], qkv[:, nq:nq + nk].float(), atol=3e-2, rtol=2e-2)
    _cmp("grouped v", Vt[:, :, :, start].reshape(M, -1), qkv[:, nq + nk:].float(), atol=3e-2, rtol=2e-2)
    Kc[:, start] = 0
    Vt[:, :, :, start] = 0
    assert float(Kc.float().abs().sum()) == 0.0 and float(Vt.float().abs().sum()) == 0.0, "only the new cache slot may be written"
    for K_ in (nq, I):
        a = _rand((M, K_), dev, 1.0, 5).to(BF)
        wl = _rand((H, K_), dev, 0.02, 6).to(BF)
        out = torch.empty((M, H), device=dev, dtype=BF)
        _lib.call("afk_decode_chain_linear_residual_batched", a.data_ptr(), K_, M, wl.data_ptr(), wl.stride(0), H, K_, res.data_ptr(), H, out.data_ptr(), H, st)
        _cmp(f"grouped linear+residual K={K_}", out, ops.gemm_nt(a, wl, residual=res).float(), atol=3e-2, rtol=2e-2)
        tail = torch.empty((M - 8, H), device=dev, dtype=BF)   # the sequences behind the first group on their own: the same sums
        _lib.call("afk_decode_chain_linear_residual_batched", a[8:].data_ptr(), K_, min(M - 8, 8), wl.data_ptr(), wl.stride(0), H, K_, res[8:].data_ptr(), H, tail.data_ptr(), H, st)
        _cmp("grouped vs eight-sequence launch", out[8:8 + min(M - 8, 8)], tail[:min(M - 8, 8)].float(), atol=2e-2, rtol=1e-2)
    wgu = _rand((2 * I, H), dev, 0.02, 7).to(BF)
    act = torch.empty((M, I), device=dev, dtype=BF)
    _lib.call("afk_decode_chain_gate_up_batched", h.data_ptr(), H, M, wgu.data_ptr(), wgu.stride(0), I, H, act.data_ptr(), I, st)
    _cmp("grouped gate|up", act, ops.silu_mul_fwd(ops.gemm_nt(h, wgu)).float(), atol=3e-2, rtol=3e-2)
    V = 4096
    wh = _rand((V, H), dev, 0.02, 8).to(BF)
    logits = torch.empty((M, V), device=dev, dtype=torch.float32)
    _lib.call("afk_decode_chain_lm_head_batched", h.data_ptr(), H, M, wh.data_ptr(), wh.stride(0), V, H, logits.data_ptr(), V, st)
    assert torch.equal(logits, logits.to(BF).float())
    _cmp("grouped lm_head", logits, ops.gemm_nt(h, wh).float(), atol=3e-2, rtol=2e-2)


@pytest.mark.parametrize("M", [1, 4, 8])
def test_decode_chain_linear_residual_norm_fused(dev, M):
    """afk_decode_chain_linear_residual_norm_batched (Linear + residual + the RMSNorm that follows in one launch; the last block to arrive normalises) against
    the two-launch sequence at the AF3-7B widths: the residual stream bit-equal to the plain batched launch of the same form, the normalised rows equal to
    afk_rmsnorm_fwd of it up to the fp32 summation order of the statistic; repeated launches (the counter resets itself), rows of different sequences do not mix."""
    from audio_flamingo_amd import _lib
    ops = _ops()
    H, nq, I = 3584, 3584, 18944
    st = ops._stream()
    cnt = torch.zeros(1, device=dev, dtype=torch.int32)
    res = _rand((M, H), dev, 1.0, 2).to(BF)
    nw = (1 + 0.1 * _rand((H,), dev, seed=3)).to(BF)
    for K_ in (nq, I):
        a = _rand((M, K_), dev, 1.0, 5).to(BF)
        wl = _rand((H, K_), dev, 0.02, 6).to(BF)
        want = torch.empty((M, H), device=dev, dtype=BF)
        _lib.call("afk_decode_chain_linear_residual_batched", a.data_ptr(), K_, M, wl.data_ptr(), wl.stride(0), H, K_, res.data_ptr(), H, want.data_ptr(), H, st)
        h_want, _ = ops.rmsnorm_fwd(want, nw, 1e-6)
        for rep in range(3):
            out = torch.full((M, H), float("nan"), device=dev, dtype=BF)
            h = torch.full((M, H), float("nan"), device=dev, dtype=BF)
            _lib.call("afk_decode_chain_linear_residual_norm_batched", a.data_ptr(), K_, M, wl.data_ptr(), wl.stride(0), H, K_, res.data_ptr(), H, out.data_ptr(), H,
                      nw.data_ptr(), 1e-6, h.data_ptr(), H, cnt.data_ptr(), st)
            torch.cuda.synchronize()
            assert int(cnt) == 0, "the hand-over counter must reset itself"
            _cmp(f"fused linear+residual K={K_}", out, want.float(), atol=3e-2, rtol=2e-2)
            if M >= 4:   # the plain launch takes the same matrix-pipe form from four sequences on
                assert torch.equal(out, want)
            ref, _ = ops.rmsnorm_fwd(out, nw, 1e-6)
            neq = int((h != ref).sum())
            assert neq <= M * H // 200, f"fused norm: {neq} of {M * H} values differ from afk_rmsnorm_fwd of the same rows"
            _cmp(f"fused norm K={K_}", h, ref.float(), atol=2e-2, rtol=1e-2)
        solo_o = torch.empty((1, H), device=dev, dtype=BF)
        solo_h = torch.empty((1, H), device=dev, dtype=BF)
        _lib.call("afk_decode_chain_linear_residual_norm_batched", a[M - 1:].data_ptr(), K_, 1, wl.data_ptr(), wl.stride(0), H, K_, res[M - 1:].data_ptr(), H, solo_o.data_ptr(), H,
                  nw.data_ptr(), 1e-6, solo_h.data_ptr(), H, cnt.data_ptr(), st)
        assert torch.equal(solo_o[0], out[M - 1]) and torch.equal(solo_h[0], h[M - 1])


@pytest.mark.parametrize("rows,D", [(1500, 1280), (12000, 1280), (37, 64)])
def test_layernorm_bwd_colsum_and_gelu_bwd_colsum(dev, rows, D):
    """bias gradients where their operand is produced: afk_layernorm_bwd_colsum / afk_gelu_bwd_colsum write the same dx (bit for bit) and the same dw / db as the
    plain entry points, and the column sums of that dx - fresh and accumulated - equal a column-sum pass over it up to the fp32 summation order"""
    ops = _ops()
    x = _rand((rows, D), dev, 1.0, 1).to(BF)
    w = (1 + 0.1 * _rand((D,), dev, seed=2)).to(BF)
    b = (0.1 * _rand((D,), dev, seed=3)).to(BF)
    dy = _rand((rows, D), dev, 1.0, 4).to(BF)
    skip = _rand((rows, D), dev, 1.0, 5).to(BF)
    _, mean, rstd = ops.layernorm_fwd(x, w, b)
    dw0, db0 = torch.zeros(D, device=dev, dtype=BF), torch.zeros(D, device=dev, dtype=BF)
    dx0 = ops.layernorm_bwd(x, w, dy, mean, rstd, dw0, db0, dx_add=skip)
    dw1, db1 = torch.zeros(D, device=dev, dtype=BF), torch.zeros(D, device=dev, dtype=BF)
    cs = torch.full((D,), float("nan"), device=dev, dtype=BF)
    dx1 = ops.layernorm_bwd(x, w, dy, mean, rstd, dw1, db1, dx_add=skip, colsum_out=cs)
    assert torch.equal(dx0, dx1) and torch.equal(dw0, dw1) and torch.equal(db0, db1)
    want = dx0.float().sum(0)
    tol = 2e-2 * want.abs() + 2e-2 * float(dx0.float().abs().sum(0).max()) / 256
    assert bool(((cs.float() - want).abs() <= tol).all()), f"column sums off by {float((cs.float() - want).abs().max())}"
    base = _rand((D,), dev, 1.0, 6).to(BF)
    acc = base.clone()
    ops.layernorm_bwd(x, w, dy, mean, rstd, dw1, db1, dx_add=skip, colsum_out=acc, colsum_accumulate=True)
    assert bool(((acc.float() - (want + base.float())).abs() <= tol + 1e-2 * base.float().abs()).all())
    # GELU backward: [rows, 4 D] as the encoder's fc1
    C = 4 * D
    pre = _rand((rows, C), dev, 1.5, 7).to(BF)
    dyg = _rand((rows, C), dev, 1.0, 8).to(BF)
    g0 = ops.gelu_bwd(dyg, pre)
    csg = torch.full((C,), float("nan"), device=dev, dtype=BF)
    g1 = ops.gelu_bwd(dyg, pre, colsum_out=csg)
    assert torch.equal(g0, g1)
    wantg = g0.float().sum(0)
    tolg = 2e-2 * wantg.abs() + 2e-2 * float(g0.float().abs().sum(0).max()) / 256
    assert bool(((csg.float() - wantg).abs() <= tolg).all()), f"GELU column sums off by {float((csg.float() - wantg).abs().max())}"
    ref = torch.empty(C, device=dev, dtype=BF)
    ops.colsum(g0, ref)
    assert int((ref != csg).sum()) <= C // 20, "the fused sums and the column-sum pass differ in more than the odd last bit"


# ------------------------------------------------------------------------------------------------ CE
def test_cross_entropy(dev):
    ops = _ops()
    rows, V = 70, 1000 + 64
    logits = _rand((rows, V), dev, 3.0, 1).to(BF)
    g = torch.Generator().manual_seed(1)
    labels = torch.randint(0, V, (rows,), generator=g)
    labels[::5] = -100
    labels = labels.to(dev)
    lr = logits.float().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lr, labels, ignore_index=-100)
    ref.backward()
    denom = ops.count_valid(labels)
    assert denom.item() == (labels >= 0).sum().item()
    row_loss = torch.empty(rows, device=dev, dtype=torch.float32)
    work = logits.clone()
    ops.ce_fwd_bwd_(work, labels, row_loss, denom)
    loss = torch.zeros(1, device=dev)
    ops.loss_reduce(row_loss, denom, loss)
    assert abs(loss.item() - ref.item()) < 2e-4 * abs(ref.item()) + 1e-5, f"CE loss {loss.item()} vs {ref.item()}"
    _cmp("dlogits", work, lr.grad, atol=2e-4, rtol=2e-2)


# ------------------------------------------------------------------------------------------------ AdamW
def test_adamw(dev):
    ops = _ops()
    n = 4099
    p = torch.nn.Parameter(_rand((n,), dev, 1.0, 1))
    opt = torch.optim.AdamW([p], lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    master = p.detach().clone()
    m, v = torch.zeros_like(master), torch.zeros_like(master)
    pb = master.to(BF)
    for step in range(1, 4):
        g = _rand((n,), dev, 1.0, 10 + step).to(BF)
        p.grad = g.float()
        opt.step()
        ops.adamw_step(master, m, v, g, pb, lr=1e-2, beta1=0.9, beta2=0.95, eps=1e-8, weight_decay=0.1, step=step, max_blocks=(3 if step == 2 else 0))
    _cmp("adamw master", master, p.detach(), atol=1e-5, rtol=1e-4)
    assert torch.equal(pb, master.to(BF))


def test_sumsq_and_clip_coef(dev):
    """global-norm clipping pieces: fp32 sum of squares of a bf16 range (odd length, gate, accumulate, deterministic) and the coefficient"""
    ops = _ops()
    x = _rand((1 << 22) + 5, dev, 1.0, 1).to(BF)
    acc = torch.zeros(1, device=dev)
    ops.sumsq_(x, acc)
    ref = x.double().pow(2).sum().item()
    assert abs(acc.item() - ref) <= 1e-5 * ref
    first = acc.clone()
    ops.sumsq_(x[:4096], acc)                                   # accumulates
    assert abs(acc.item() - (ref + x[:4096].double().pow(2).sum().item())) <= 1e-5 * ref
    ops.sumsq_(x, acc, gate=torch.zeros(1, device=dev, dtype=torch.int32))   # gated off on the device
    again = torch.zeros(1, device=dev)
    ops.sumsq_(x, again)
    assert torch.equal(again, first), "sum of squares not deterministic"
    coef, norm = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    parts = torch.cat([first * 0.25, first * 0.75])            # per-bucket slots are folded in index order
    ops.clip_coef_(parts, coef, max_norm=1.0, scale=0.5, norm_out=norm)
    n = 0.5 * ref ** 0.5
    assert abs(norm.item() - n) <= 1e-5 * n and abs(coef.item() - 1.0 / (n + 1e-6)) <= 1e-6
    ops.clip_coef_(first, coef, max_norm=1e9)
    assert coef.item() == 1.0


# ------------------------------------------------------------------------------------------------ log-mel
def test_logmel(dev):
    """vs the reference feature extractor itself (transformers WhisperFeatureExtractor, feature_extraction_whisper.py:135-168 numpy path and
    :170-201 torch path, both on the CPU).  The reference's two paths do not agree to better than ~3e-5 on this input (log10 of near-floor
    bins amplifies fp32 rounding), so the bar is stated against that: ours within 4x the reference's own path-to-path deviation of either
    path, and a median below 1e-5."""
    import numpy as np
    from transformers import WhisperFeatureExtractor
    from audio_flamingo_amd.frontend import LogMelFrontend, mel_filter_bank

    rng = np.random.default_rng(0)
    n = 480000
    wav = np.zeros((2, n), np.float32)
    wav[0] = rng.standard_normal(n).astype(np.float32) * 0.1
    t = np.arange(80000) / 16000.0
    wav[1, :80000] = (0.3 * np.sin(2 * np.pi * 440 * t) + 0.01 * rng.standard_normal(80000)).astype(np.float32)  # 5 s clip, zero padded
    ref_fe = WhisperFeatureExtractor(feature_size=128, sampling_rate=16000, hop_length=160, chunk_length=30, n_fft=400)
    assert np.array_equal(mel_filter_bank(128), ref_fe.mel_filters)  # the filter bank is the reference's, bit for bit
    ref_np = torch.from_numpy(np.asarray(ref_fe._np_extract_fbank_features(wav, "cpu"), np.float32))
    ref_t = torch.from_numpy(np.asarray(ref_fe._torch_extract_fbank_features(wav, "cpu"), np.float32))
    d_ref = (ref_np - ref_t).abs().max().item()
    w = torch.from_numpy(wav)
    fe = LogMelFrontend(dev)
    out = fe(w.to(dev))
    assert out.shape == (2, 128, 3000)
    e_t, e_np = (out.cpu() - ref_t).abs(), (out.cpu() - ref_np).abs()
    err = torch.minimum(e_t, e_np)
    bar = 4.0 * max(d_ref, 1e-5)
    msg = f"logmel: ours vs torch path {e_t.max().item():.3g}, vs numpy path {e_np.max().item():.3g}, reference's own two paths {d_ref:.3g}"
    print(msg)
    assert err.max().item() <= bar, msg
    assert e_t.median().item() < 1e-5, f"logmel median err {e_t.median().item()}"
    outb = fe(w.to(dev), out_dtype=torch.bfloat16)
    assert torch.equal(outb, out.to(torch.bfloat16))
