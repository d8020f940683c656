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
