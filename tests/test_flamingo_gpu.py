"""BASELINE config 4 (AF1/AF2-style Perceiver resampler + tanh-gated cross-attention, 4 clips per sample) on MI355X against the
CPU oracle restatement of the structural stand-in (oracle/flamingo_oracle.py; parity w.r.t. AF1/AF2 itself is UNPINNED)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def _bf_state(mod, seed):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in mod.state_dict().items():
        if k.endswith("norm.weight") or k.endswith("ln.weight"):
            t = 1 + 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith(".bias"):
            t = 0.05 * torch.randn(v.shape, generator=g)
        elif k.startswith("alpha"):
            t = 0.5 * torch.randn(v.shape, generator=g)
        elif k == "latents":
            t = torch.randn(v.shape, generator=g)
        else:
            t = 0.05 * torch.randn(v.shape, generator=g)
        sd[k] = t.to(BF)
    return sd


def test_perceiver_resampler(dev):
    from audio_flamingo_amd.flamingo import PerceiverResampler
    from oracle import flamingo_oracle as FO

    E, depth, H, D, L = 128, 2, 4, 32, 64
    m = PerceiverResampler(E, depth, H, D, L, device=dev)
    sd = _bf_state(m, 0)
    m.load_state_dict(sd)
    B, T = 8, 100  # 2 samples x 4 clips, T_enc = 100
    ctx = (torch.randn(B, T, E, generator=torch.Generator().manual_seed(1))).to(BF)
    sdf = {k: v.float().requires_grad_(True) for k, v in sd.items()}
    cf = ctx.float().requires_grad_(True)
    ref = FO.perceiver_resampler(sdf, cf, H, D)
    w = torch.randn(ref.shape, generator=torch.Generator().manual_seed(2))
    (ref * w).sum().backward()
    cg = ctx.to(dev).requires_grad_(True)
    out = m(cg)
    (out.float() * w.to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert _rel(out, ref) < 2e-2, _rel(out, ref)
    assert _rel(cg.grad, cf.grad) < 5e-2, _rel(cg.grad, cf.grad)
    params = dict(m.named_parameters())
    for k in ("latents", "blocks.0.0.k_proj.weight", "blocks.1.1.fc.weight", "blocks.0.0.context_layer_norm.weight", "layer_norm.bias",
              "blocks.1.0.output_proj.weight"):
        assert _rel(params[k].grad, sdf[k].grad) < 6e-2, (k, _rel(params[k].grad, sdf[k].grad))


@pytest.mark.parametrize("alpha_type", ["vector", "float"])
def test_gated_cross_attention_block(dev, alpha_type):
    from audio_flamingo_amd.flamingo import GatedCrossAttentionBlock, media_key_ranges
    from oracle import flamingo_oracle as FO

    Hd, heads, inter, L = 256, 4, 512, 64
    blk = GatedCrossAttentionBlock(Hd, heads, inter, eps=1e-6, alpha_type=alpha_type, device=dev)
    sd = _bf_state(blk, 3)
    blk.load_state_dict(sd)
    B, S = 2, 200
    marks = [[7, 60, 110, 150], [0, 33, 90, 191]]  # 4 <audio> markers per sample; sample 0 has 7 leading tokens with no media
    kr, gate = media_key_ranges(marks, S, L)
    Sk = 4 * L
    x = torch.randn(B, S, Hd, generator=torch.Generator().manual_seed(4)).to(BF)
    media = torch.randn(B, Sk, Hd, generator=torch.Generator().manual_seed(5)).to(BF)
    keep = torch.arange(Sk)[None, None, :]
    keep = (keep >= kr[..., 0:1]) & (keep < kr[..., 1:2])
    sdf = {k: v.float().requires_grad_(True) for k, v in sd.items()}
    xf, mf = x.float().requires_grad_(True), media.float().requires_grad_(True)
    ref = FO.gated_cross_attention(sdf, xf, mf, keep, gate, heads, 1e-6)
    w = torch.randn(ref.shape, generator=torch.Generator().manual_seed(6))
    (ref * w).sum().backward()
    xg, mg = x.to(dev).requires_grad_(True), media.to(dev).requires_grad_(True)
    out = blk(xg, mg, kr.to(dev), gate.to(dev))
    (out.float() * w.to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert _rel(out, ref) < 1e-2, _rel(out, ref)
    assert _rel(xg.grad, xf.grad) < 3e-2 and _rel(mg.grad, mf.grad) < 5e-2, (_rel(xg.grad, xf.grad), _rel(mg.grad, mf.grad))
    params = dict(blk.named_parameters())
    for k in ("alpha_cross_attn", "alpha_dense", "cross_attn.k_proj.weight", "cross_attn.o_proj.weight", "mlp.up_proj.weight",
              "input_layernorm.weight"):
        if alpha_type == "float" and k.startswith("alpha"):
            # a scalar gate gradient is one heavily cancelling sum over B*S*H products (|sum| ~ 3 against sum|terms| ~ 1e4): it is
            # compared on the scale of its terms, i.e. with an absolute allowance, not as a relative error of the residue
            got, ref = float(params[k].grad.float().cpu()), float(sdf[k].grad)
            assert abs(got - ref) < 0.05 * abs(ref) + 1.0, (k, got, ref)
            continue
        assert _rel(params[k].grad, sdf[k].grad) < 6e-2, (k, _rel(params[k].grad, sdf[k].grad))
    # tokens before the first clip see no media: the cross-attention branch must contribute exactly nothing there (M:792)
    o2 = blk(x.to(dev), torch.zeros_like(media).to(dev), kr.to(dev), gate.to(dev))
    assert torch.isfinite(o2.float()).all()


def test_icl_step_assembled_vs_oracle(dev):
    """config 4 assembled: 4 clips per sample -> projection -> Perceiver resampler -> decoder with a gated cross-attention block every 2nd
    layer -> lm_head + CE; forward, every parameter gradient and one fused-AdamW step against oracle/flamingo_oracle.py::icl_forward
    (parity w.r.t. AF1/AF2 itself: UNPINNED)"""
    from audio_flamingo_amd.flamingo_icl import FlamingoICLForCausalLM, TensorAdamW
    from oracle import flamingo_oracle as FO

    c = dict(vocab=512, hidden=256, inter=512, layers=4, heads=4, kv_heads=2, head_dim=64, rms_eps=1e-6, rope_theta=1e4, xattn_every=2,
             xattn_heads=4, xattn_inter=512, n_latents=64, resampler_depth=2, resampler_heads=4, resampler_head_dim=64, enc_dim=128, enc_frames=48,
             clips=4, audio_marker_id=511)
    m = FlamingoICLForCausalLM(c, device=dev, seed=1)
    sd = _bf_state(m, 9)
    for k in list(sd):
        if "alpha" in k:
            sd[k] = (0.5 * torch.randn(sd[k].shape, generator=torch.Generator().manual_seed(len(k)))).to(BF)  # open gates
    m.load_state_dict(sd)
    B, S = 2, 192
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, 500, (B, S), generator=g)
    for b, marks in enumerate([[5, 40, 90, 130], [0, 31, 77, 160]]):
        ids[b, marks] = 511
    labels = ids.clone()
    labels[:, :100] = -100
    feats = torch.randn(B, 4, 48, 128, generator=g).to(BF)
    sdf = {k: v.float().requires_grad_(True) for k, v in sd.items()}
    ref_loss, ref_logits = FO.icl_forward(sdf, c, ids, feats.float(), labels)
    ref_loss.backward()
    shift = torch.nn.functional.pad(labels, (0, 1), value=-100)[:, 1:].reshape(-1)
    rows = (shift != -100).nonzero().reshape(-1).to(dev)
    loss = m(ids.to(dev), feats.to(dev), labels.to(dev), label_rows=rows)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(ref_loss)) <= 1e-2, (float(loss), float(ref_loss))
    with torch.no_grad():
        lg = m(ids.to(dev), feats.to(dev)).float().cpu()
    assert float((lg - ref_logits.detach()).abs().max()) <= 4e-2 * max(1.0, float(ref_logits.abs().max()))
    params = dict(m.named_parameters())
    bad = {k: _rel(params[k].grad, v.grad) for k, v in sdf.items() if v.grad is not None and v.grad.norm() > 0 and _rel(params[k].grad, v.grad) > 8e-2}
    assert not bad, bad
    opt = TensorAdamW(m.parameters(), lr=1e-3)
    before = float(loss)
    for _ in range(3):
        opt.step()
        opt.zero_grad()
        loss = m(ids.to(dev), feats.to(dev), labels.to(dev), label_rows=rows)
        loss.backward()
    assert float(loss) < before, (before, float(loss))
