"""torch.library registration of the kernels (audio_flamingo_amd/custom_ops.py) - what can be checked without a GPU: the ops exist under
torch.ops.afk with the declared schemas, their shape functions run under FakeTensorMode on (fake) HIP tensors, a decoder layer built from them
traces into ONE graph whose nodes are the registered ops, and a CPU tensor reaches no kernel (no CPU fallback)."""
import pytest
import torch


def _layer(C, B, S, Hq, Hkv, D):
    def layer(x, wn1, wqkv, bqkv, wo, wn2, wgu, wd, cos, sin):
        h = C.rms_norm(x, wn1, 1e-6)
        qkv = torch.ops.afk.linear(h, wqkv, bqkv)
        qkv = torch.ops.afk.rope(qkv, cos, sin, S, Hq + Hkv, D, False)
        o = C.attention(qkv, B, S, Hq, Hkv, D)
        x2 = torch.ops.afk.linear(o, wo, None) + x
        h2 = C.rms_norm(x2, wn2, 1e-6)
        a = torch.ops.afk.silu_mul(torch.ops.afk.linear(h2, wgu, None))
        return torch.ops.afk.linear(a, wd, None) + x2

    return layer


def layer_args(B, S, Hq, Hkv, D, H, I, device, fill=None):
    mk = (lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=device)) if fill is None else fill
    return [mk(B * S, H), mk(H), mk((Hq + 2 * Hkv) * D, H), mk((Hq + 2 * Hkv) * D), mk(H, Hq * D), mk(H), mk(2 * I, H), mk(H, I), mk(S, D), mk(S, D)]


def test_ops_are_registered_with_schemas():
    from audio_flamingo_amd import custom_ops as C

    for name in C.REGISTERED:
        op = getattr(torch.ops.afk, name)
        assert str(op.default._schema).startswith(f"afk::{name}("), op.default._schema
    assert str(torch.ops.afk.linear.default._schema) == "afk::linear(Tensor x, Tensor weight, Tensor? bias=None) -> Tensor"
    with pytest.raises(NotImplementedError, match="CPU"):   # registered for the HIP device only: nothing computes on the host
        torch.ops.afk.linear(torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(8, 64, dtype=torch.bfloat16), None)


def test_shape_functions_and_single_graph_trace():
    from torch._subclasses.fake_tensor import FakeTensorMode
    from torch.fx.experimental.proxy_tensor import make_fx

    from audio_flamingo_amd import custom_ops as C

    B, S, Hq, Hkv, D, H, I = 2, 100, 4, 2, 64, 256, 512
    with FakeTensorMode(allow_non_fake_inputs=True):
        args = layer_args(B, S, Hq, Hkv, D, H, I, "cuda")
        o, lse = torch.ops.afk.attention_fwd(args[0].new_empty(B * S, (Hq + 2 * Hkv) * D), B, S, Hq, Hkv, D, 0.125, True)
        assert o.shape == (B * S, Hq * D) and lse.shape == (B, Hq, 128) and lse.dtype == torch.float32 and o.device.type == "cuda"
        dx, dw, db = torch.ops.afk.linear_bwd(args[0].new_empty(B * S, 384), args[0], args[0].new_empty(384, H), True)
        assert dx.shape == (B * S, H) and dw.shape == (384, H) and db.shape == (384,)
        gm = make_fx(_layer(C, B, S, Hq, Hkv, D), tracing_mode="fake")(*args)
    targets = [str(n.target) for n in gm.graph.nodes if n.op == "call_function"]
    afk = [t for t in targets if t.startswith("afk.")]
    assert afk == ["afk.rms_norm_fwd.default", "afk.linear.default", "afk.rope.default", "afk.attention_fwd.default", "afk.linear.default",
                   "afk.rms_norm_fwd.default", "afk.linear.default", "afk.silu_mul.default", "afk.linear.default"], afk


def test_stage_operator_plumbing_with_a_toy_stage():
    """stage_ops.register_stage (the dispatcher plumbing every stage of the training step runs on since round 4) on a toy stage whose body is
    plain torch: forward / backward operators with schema, the gradient arena written in place by the BACKWARD operator (declared mutated),
    optional inputs, an input and a None among the tensors kept for backward, activation checkpointing, no-grad calls, and
    torch.compile(fullgraph=True) through the registered operators."""
    import torch
    from torch.utils.checkpoint import checkpoint

    from audio_flamingo_amd import stage_ops

    class Arena:   # what a stage sees of the arena: flat parameter / gradient buffers + named views
        def __init__(self):
            self.params = torch.arange(1, 7, dtype=torch.float32).reshape(-1) / 10
            self.grads = torch.zeros(6)
            self.w = self.params[:6].view(2, 3)
            self.gw = self.grads[:6].view(2, 3)

    class ToyFn:
        @staticmethod
        def forward(ctx, x, anchor, arena, scale, bias):
            h = x @ arena.w.t()                      # [n, 2]
            y = torch.tanh(h) * scale
            if bias is not None:
                y = y + bias
            ctx.save_for_backward(x, h, None, bias)  # an input, an activation, a None, an optional input
            ctx.meta = (arena, scale)
            return y

        @staticmethod
        def backward(ctx, dy):
            x, h, nothing, bias = ctx.saved_tensors
            arena, scale = ctx.meta
            assert nothing is None
            dh = dy * scale * (1 - torch.tanh(h) ** 2)
            arena.gw += dh.t() @ x                   # weight gradient straight into the gradient arena
            return dh @ arena.w, None, None, None, (dy.sum(0) if bias is not None else None)

    name = "toy_stage_for_the_cpu_test"
    if name not in stage_ops.registered_stages():
        ToyFn.apply = staticmethod(stage_ops.register_stage(name, ToyFn, ("T", "T", "A", "S", "T?")))
    else:
        ToyFn.apply = staticmethod(stage_ops._STAGES[name].apply)
    assert "Tensor(a!) grads" in str(getattr(torch.ops.afk, name + "_bwd").default._schema)
    assert "!" not in str(getattr(torch.ops.afk, name).default._schema)

    def reference(x, w, bias, scale):
        y = torch.tanh(x @ w.t()) * scale
        return y + bias if bias is not None else y

    for use_bias in (True, False):
        a = Arena()
        anchor = torch.nn.Parameter(torch.zeros(1))
        x = torch.randn(5, 3, requires_grad=True)
        b = torch.randn(2, requires_grad=True) if use_bias else None
        y = ToyFn.apply(x, anchor, a, 2.0, b)
        y.square().sum().backward()
        w = a.w.clone().requires_grad_(True)
        xr = x.detach().clone().requires_grad_(True)
        br = b.detach().clone().requires_grad_(True) if use_bias else None
        reference(xr, w, br, 2.0).square().sum().backward()
        assert torch.allclose(x.grad, xr.grad, atol=1e-6) and torch.allclose(a.gw, w.grad, atol=1e-6)
        assert anchor.grad is None
        if use_bias:
            assert torch.allclose(b.grad, br.grad, atol=1e-6)
        # checkpointed: same gradients (the backward operator runs once, on the recomputed activations)
        a2 = Arena()
        x2 = x.detach().clone().requires_grad_(True)
        y2 = checkpoint(ToyFn.apply, x2, anchor, a2, 2.0, b, use_reentrant=False)
        y2.square().sum().backward()
        assert torch.allclose(x2.grad, xr.grad, atol=1e-6) and torch.allclose(a2.gw, w.grad, atol=1e-6)
        # no-grad call: forward only, nothing recorded
        with torch.no_grad():
            assert torch.equal(ToyFn.apply(x, anchor, a, 2.0, b), y.detach())

    # the whole toy step under torch.compile(fullgraph=True): the operators are opaque graph nodes, no graph break
    a3 = Arena()
    anchor = torch.nn.Parameter(torch.zeros(1))
    x3 = torch.randn(5, 3, requires_grad=True)
    b3 = torch.randn(2, requires_grad=True)

    def step(x, b):
        return ToyFn.apply(ToyFn.apply(x, anchor, a3, 2.0, b)[:, :2] @ torch.ones(2, 3), anchor, a3, 2.0, b).square().sum()

    eager = step(x3, b3)
    eager.backward()           # also records the stage geometry the fake implementations answer from
    g_eager, gw_eager = x3.grad.clone(), a3.gw.clone()
    x3.grad = None
    a3.grads.zero_()
    compiled = torch.compile(step, fullgraph=True, backend="aot_eager")
    out = compiled(x3, b3)
    out.backward()
    assert torch.allclose(out, eager) and torch.allclose(x3.grad, g_eager, atol=1e-6) and torch.allclose(a3.gw, gw_eager, atol=1e-6)
    # a traced step bakes its stage keys in as constants and never goes through _Stage.apply() again - traced keys are pinned, so the LRU cap of
    # the side tables evicts only untraced geometries, and the compiled step still runs after more than _TABLE_CAP other geometries have passed through
    traced = [k for k in stage_ops._PINNED if k.startswith(name + "|")]
    assert traced and all(k in stage_ops._CACHE and k in stage_ops._STATIC for k in traced)
    cap = stage_ops._TABLE_CAP
    try:
        stage_ops._TABLE_CAP = len(stage_ops._STATIC) + 2
        for n in range(6, 14):   # eight more geometries: more than the cap leaves room for
            with torch.no_grad():
                ToyFn.apply(torch.randn(n, 3), anchor, a3, 2.0, None)
        assert len(stage_ops._STATIC) <= stage_ops._TABLE_CAP + len(stage_ops._PINNED)
        assert all(k in stage_ops._CACHE and k in stage_ops._STATIC for k in traced)
        x3.grad = None
        a3.grads.zero_()
        out2 = compiled(x3, b3)
        out2.backward()
        assert torch.allclose(out2, eager) and torch.allclose(x3.grad, g_eager, atol=1e-6)
    finally:
        stage_ops._TABLE_CAP = cap
    # and an evicted (never traced) key fails with the explicit message, not a bare KeyError
    import pytest as _pt

    with _pt.raises(RuntimeError, match="evicted"):
        stage_ops._cache_of("no_such_stage|0|()|()|g0")
