"""The registered operators (torch.ops.afk.*, audio_flamingo_amd/custom_ops.py) on the MI355X: torch.library.opcheck (schema, fake tensor,
autograd registration, AOT dispatch), and a Qwen2 decoder layer built from them - eager AND under torch.compile (one graph, no breaks) -
against the layer-level stage of the training step (functional.DecoderLayerFn), forward and every gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _rand(shape, dev, scale, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev).to(BF)


def test_opcheck(dev):
    from audio_flamingo_amd import custom_ops as C  # noqa: F401

    x = _rand((200, 128), dev, 1.0, 1).requires_grad_()
    w = _rand((256, 128), dev, 0.1, 2).requires_grad_()
    b = _rand((256,), dev, 0.1, 3).requires_grad_()
    tests = ("test_schema", "test_faketensor", "test_autograd_registration", "test_aot_dispatch_dynamic")
    torch.library.opcheck(torch.ops.afk.linear.default, (x, w, b), test_utils=tests)
    torch.library.opcheck(torch.ops.afk.linear.default, (x, w, None), test_utils=tests)
    g = _rand((128,), dev, 1.0, 4).requires_grad_()
    torch.library.opcheck(torch.ops.afk.rms_norm_fwd.default, (x, g, 1e-6), test_utils=tests)
    torch.library.opcheck(torch.ops.afk.layer_norm_fwd.default, (x, g, _rand((128,), dev, 0.1, 5).requires_grad_(), 1e-5), test_utils=tests)
    qkv = _rand((2 * 100, (4 + 4) * 64), dev, 1.0, 6).requires_grad_()
    torch.library.opcheck(torch.ops.afk.attention_fwd.default, (qkv, 2, 100, 4, 2, 64, 0.125, True), test_utils=tests)
    torch.library.opcheck(torch.ops.afk.silu_mul.default, (_rand((64, 256), dev, 1.0, 7).requires_grad_(),), test_utils=tests)
    torch.library.opcheck(torch.ops.afk.gelu.default, (_rand((64, 256), dev, 1.0, 8).requires_grad_(),), test_utils=tests)


@pytest.mark.parametrize("compiled", [False, True])
def test_decoder_layer_from_registered_ops_matches_the_training_stage(dev, compiled):
    """same kernels, two hosts: the layer-level autograd stage of the training step (weight gradients written into the arena) and a layer
    composed from torch.ops.afk.* - forward bit-identical; gradients agree to bf16 rounding (the registered linear_bwd runs dgrad on the NN
    kernel from W as stored, the stage on the NT kernel from the W^T shadow: different summation order).  compiled: the composed layer runs
    under torch.compile(fullgraph=True, backend="aot_eager") - the ctypes calls are opaque graph nodes, no graph break"""
    from audio_flamingo_amd import custom_ops as C
    from audio_flamingo_amd import functional as F_
    from audio_flamingo_amd.arena import Arena
    from tests.test_custom_ops_cpu import _layer

    B, S, Hq, Hkv, D, H, I = 2, 256, 4, 2, 64, 256, 512
    nq, nkv = Hq * D, Hkv * D
    a = Arena(dev)
    bk = a.new_bucket("l")
    pfx = "l."
    a.add(pfx + "input_layernorm.weight", (H,), bk, decay=False)
    a.add(pfx + "self_attn.qkv.weight", (nq + 2 * nkv, H), bk, shadow="T")
    a.add(pfx + "self_attn.qkv.bias", (nq + 2 * nkv,), bk, decay=False)
    a.add(pfx + "self_attn.o_proj.weight", (H, nq), bk, shadow="T")
    a.add(pfx + "post_attention_layernorm.weight", (H,), bk, decay=False)
    a.add(pfx + "mlp.gate_up.weight", (2 * I, H), bk, shadow="T")
    a.add(pfx + "mlp.down_proj.weight", (H, I), bk, shadow="T")
    a.finalize()
    for i, blk in enumerate(a.order):
        blk.data.copy_(_rand(blk.shape, dev, 0.05, 10 + i) + (1.0 if "layernorm" in blk.key else 0.0))
    a.refresh_shadows(force=True)
    x0 = _rand((B * S, H), dev, 1.0, 1)
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    fr = torch.arange(S, dtype=torch.float32)[:, None] * inv[None]
    emb = torch.cat([fr, fr], -1)
    cos, sin = emb.cos().to(dev).to(BF).contiguous(), emb.sin().to(dev).to(BF).contiguous()
    up = _rand((B * S, H), dev, 1.0, 2)
    # (i) the training stage
    xs = x0.clone().requires_grad_()
    anchor = torch.nn.Parameter(a[pfx + "mlp.down_proj.weight"].data)
    a.zero_grad()
    ys = F_.DecoderLayerFn.apply(xs, anchor, a, pfx, B, S, Hq, Hkv, D, 1e-6, cos, sin, None, None, None, None)
    ys.backward(up)
    # (ii) the composed layer
    names = ["input_layernorm.weight", "self_attn.qkv.weight", "self_attn.qkv.bias", "self_attn.o_proj.weight", "post_attention_layernorm.weight",
             "mlp.gate_up.weight", "mlp.down_proj.weight"]
    prm = [a[pfx + n].data.clone().requires_grad_() for n in names]
    xc = x0.clone().requires_grad_()
    fn = _layer(C, B, S, Hq, Hkv, D)
    if compiled:
        torch._dynamo.reset()
        fn = torch.compile(fn, fullgraph=True, backend="aot_eager")
    yc = fn(xc, *prm, cos, sin)
    yc.backward(up)
    torch.cuda.synchronize()

    def rel(u, v):
        return float((u.float() - v.float()).norm() / v.float().norm().clamp_min(1e-20))

    # forward: every stage except the MLP is the same launch; the stage fuses SwiGLU into the gate|up GEMM (bit-identical by test_gemm_swiglu_fwd_epilogue)
    # and the residual adds into the GEMM epilogues (round-to-bf16 then add: the same two roundings as linear + add)
    assert rel(yc, ys) <= 2e-3, rel(yc, ys)
    assert rel(xc.grad, xs.grad) <= 2e-2, rel(xc.grad, xs.grad)
    for n, p in zip(names, prm):
        r = rel(p.grad, a[pfx + n].grad)
        assert r <= 2e-2, (n, r)
