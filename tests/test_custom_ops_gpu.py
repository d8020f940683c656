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


def test_training_step_runs_on_registered_stage_operators_and_traces_under_torch_compile(dev):
    """the product path IS the registered-operator path.  (1) One training step of the tiny AF3 model dispatches every stage
    through torch.ops.afk.* (dispatch counts per stage; the backward operators declare the gradient arena as mutated).  (2) The stage chain of that
    step - conv stem, encoder layers, pool + LayerNorm, projector, embedding scatter, decoder layers, final RMSNorm, lm_head + loss - traces into
    ONE graph under torch.compile(fullgraph=True) (aot_eager: the ctypes calls into libafk.so stay opaque operator nodes) and reproduces the eager
    loss and every gradient bit for bit."""
    import os

    from audio_flamingo_amd import functional as F_
    from audio_flamingo_amd import ops, stage_ops
    from tests.test_model_gpu import G, _model

    g = torch.load(os.path.join(G, "tiny64_caseA.pt"))
    m = _model(dev)
    m.check_placeholders = False
    m.loss_on_valid_rows_only = False      # the chain below feeds lm_head + loss every row, as the reference does: same arithmetic in both runs
    ids, feats, labels = g["ids"].to(dev), g["feats"].to(dev), g["labels"].to(dev)

    # ---- (1) the model's own forward / backward: count dispatches per registered stage
    counts = {}
    orig = {}
    for name, st in stage_ops._STAGES.items():
        orig[name] = (st.op, st.op_bwd)

        def wrap(op, nm):
            def call(*a):
                counts[nm] = counts.get(nm, 0) + 1
                return op(*a)
            return call
        st.op, st.op_bwd = wrap(st.op, name), wrap(st.op_bwd, name + "_bwd")
    try:
        m.zero_grad()
        out = m(input_ids=ids, input_features=feats, labels=labels)
        out.loss.backward()
        torch.cuda.synchronize()
    finally:
        for name, st in stage_ops._STAGES.items():
            st.op, st.op_bwd = orig[name]
    L_enc, L_dec = m.enc_layers, m.dec_layers
    want = {"conv_stem": 1, "encoder_layer": L_enc, "pool_norm": 1, "projector": 1, "embed_scatter": 1, "decoder_layer": L_dec, "final_rms_norm": 1, "lm_head_loss": 1}
    for k, v in want.items():
        assert counts.get(k) == v and counts.get(k + "_bwd") == v, (k, counts)
    for name in want:
        assert "Tensor(a!) grads" in str(getattr(torch.ops.afk, name + "_bwd").default._schema)
    eager_loss, eager_grads = float(out.loss), m.arena.grads.clone()

    # ---- (2) the same stage chain as one traced graph
    a, at, lm, pj = m.arena, m._at, m._lm, m._pj
    W, C, T = feats.shape
    T2 = (T - 1) // 2 + 1
    T3 = T2 // 2
    B, S = ids.shape
    ids_flat = ids.reshape(-1).contiguous()
    src, _ = ops.placeholder_scan(ids_flat, m.audio_token_id)
    cos, sin = m._rope_tables(S)
    shift = torch.nn.functional.pad(labels, (0, 1), value=-100)[:, 1:].reshape(-1).contiguous()
    denom = ops.count_valid(shift)
    anchors = {k: m._anchor(k) for k in [at + "conv1.weight", at + "layer_norm.weight", pj + "linear_1.weight", lm + "embed_tokens.weight", lm + "norm.weight", "lm_head.weight"]
               + [f"{at}layers.{i}.fc1.weight" for i in range(L_enc)] + [f"{lm}layers.{i}.mlp.down_proj.weight" for i in range(L_dec)]}
    pos_table = m.embed_positions.data

    def chain(feats_, ids_flat_, src_, shift_, denom_):
        x = F_.ConvStemFn.apply(feats_, anchors[at + "conv1.weight"], a, (at + "conv1.weight", at + "conv1.bias", at + "conv2.weight", at + "conv2.bias"), pos_table, W, T, C)
        for i in range(L_enc):
            p = f"{at}layers.{i}."
            x = F_.EncoderLayerFn.apply(x, anchors[p + "fc1.weight"], a, p, W, T2, m.enc_heads, None)
        x = F_.PoolNormFn.apply(x, anchors[at + "layer_norm.weight"], a, at + "layer_norm.weight", at + "layer_norm.bias", W * T3)
        x = F_.ProjectorFn.apply(x, anchors[pj + "linear_1.weight"], a, pj)
        x = F_.EmbedScatterFn.apply(x, anchors[lm + "embed_tokens.weight"], a, lm + "embed_tokens.weight", ids_flat_, src_)
        for i in range(L_dec):
            p = f"{lm}layers.{i}."
            x = F_.DecoderLayerFn.apply(x, anchors[p + "mlp.down_proj.weight"], a, p, B, S, m.Hq, m.Hkv, m.D, m.rms_eps, cos, sin, None, None, None, None)
        x = F_.RMSNormFn.apply(x, anchors[lm + "norm.weight"], a, lm + "norm.weight", m.rms_eps)
        return F_.LMHeadLossFn.apply(x, anchors["lm_head.weight"], a, "lm_head.weight", shift_, denom_, None)

    m.zero_grad()
    l0 = chain(feats, ids_flat, src, shift, denom)     # eager pass of the chain: also records the stage geometries the fake implementations answer from
    l0.backward()
    torch.cuda.synchronize()
    assert abs(float(l0) - eager_loss) < 1e-6
    assert torch.equal(m.arena.grads, eager_grads)
    m.zero_grad()
    compiled = torch.compile(chain, fullgraph=True, backend="aot_eager")
    l1 = compiled(feats, ids_flat, src, shift, denom)
    l1.backward()
    m.arena.join_streams()
    torch.cuda.synchronize()
    assert float(l1) == float(l0)
    assert torch.equal(m.arena.grads, eager_grads), "traced step: gradients differ from the eager step"


def test_logmel_and_adamw_operators(dev):
    """afk::logmel (what LogMelFrontend dispatches) and afk::adamw_step (mutating operator for hosts that step after backward) against the direct C-ABI calls"""
    from audio_flamingo_amd import custom_ops as C  # noqa: F401
    from audio_flamingo_amd import ops
    from audio_flamingo_amd.frontend import LogMelFrontend

    fe = LogMelFrontend(dev)
    g = torch.Generator().manual_seed(0)
    wav = (0.1 * torch.randn(2, 480000, generator=g)).to(dev)
    a = fe(wav)
    b = torch.ops.afk.logmel(wav, fe.cosb, fe.sinb, fe.melT, fe.n_mels, fe.nbins_pad, False)
    assert a.shape == (2, 128, 3000) and torch.equal(a, b)
    torch.library.opcheck(torch.ops.afk.logmel.default, (wav, fe.cosb, fe.sinb, fe.melT, fe.n_mels, fe.nbins_pad, True), test_utils=("test_schema", "test_faketensor"))
    n = 4096
    st = [torch.randn(n, device=dev) for _ in range(3)]
    st[2] = st[2].abs()
    grad = torch.randn(n, device=dev).to(BF)
    p1, p2 = torch.zeros(n, device=dev, dtype=BF), torch.zeros(n, device=dev, dtype=BF)
    s1, s2 = [t.clone() for t in st], [t.clone() for t in st]
    kw = dict(lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, step=3, grad_scale=0.5, max_blocks=0)
    ops.adamw_step(s1[0], s1[1], s1[2], grad, p1, **kw)
    torch.ops.afk.adamw_step(s2[0], s2[1], s2[2], grad, p2, kw["lr"], kw["beta1"], kw["beta2"], kw["eps"], kw["weight_decay"], kw["step"], kw["grad_scale"], 0)
    assert torch.equal(p1, p2) and all(torch.equal(x, y) for x, y in zip(s1, s2))
    assert "Tensor(a" in str(torch.ops.afk.adamw_step.default._schema)
