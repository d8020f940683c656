"""The reference's OWN model code (transformers AudioFlamingo3ForConditionalGeneration, run on the ROCm device) with its attention
dispatched to libafk.so through the reference's plugin registry (audio_flamingo_amd/hf_plugin.py).

  * against the golden vectors of the fp32 CPU reference (same tolerances as tests/test_model_gpu.py), forward AND backward
  * against the reference's stock `sdpa` path on the same device / weights / inputs (bf16 noise floor), incl. generate()
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def _hf_model(cfg_dict, dev, state=None, seed=0):
    from transformers import AudioFlamingo3Config, AudioFlamingo3ForConditionalGeneration

    torch.manual_seed(seed)
    m = AudioFlamingo3ForConditionalGeneration(AudioFlamingo3Config(**cfg_dict))
    if state is not None:
        m.load_state_dict(state)
    from tools.parity_fulldepth import restore_rope_buffers   # .to(bfloat16) rounds the rotary inv_freq buffer; from_pretrained(dtype=bf16) keeps it fp32

    return restore_rope_buffers(m.to(dev).to(torch.bfloat16))


@pytest.mark.parametrize("case", ["A", "B", "C"])
def test_reference_model_with_afk_attention_vs_golden(dev, case):
    """the reference's own bf16 model on this device, attention through libafk.so, against the fp32 golden - beside the SAME model with its
    stock sdpa attention (the bf16 noise floor of these trained, sharp-softmax goldens): the plugin run may deviate from the golden by the
    bars of tests/_tol.py or by the floor rule of that file (2 x the LARGEST deviation of the stock run over the floor batches of
    tests/test_model_gpu.py::_floor_distribution + this batch; gradient bars capped, noise-dominated tensors reported only), whichever is larger"""
    from audio_flamingo_amd import hf_plugin
    from tests._tol import GRAD_REL_L2, LOSS_ATOL, NOISE_DOMINATED, floor_range_bar as floor_bar, logit_tol
    from tests.test_host_cpu import TINY
    from tests.test_model_gpu import _floor_distribution

    name = hf_plugin.register()
    g = torch.load(os.path.join(G, f"tiny64_case{case}.pt"))
    m = _hf_model(TINY, dev, torch.load(os.path.join(G, "tiny64_state_bf16.pt")))
    m.train()
    att = g["att"].to(dev) if case != "A" else None
    sel = g["labels"] != -100
    res = {}
    for impl in ("sdpa", name):
        m.set_attn_implementation(impl)
        m.zero_grad()
        before = hf_plugin.calls["interval"] + hf_plugin.calls["lds"]
        out = m(input_ids=g["ids"].to(dev), input_features=g["feats"].to(dev), input_features_mask=g["fmask"].to(dev),
                attention_mask=att, labels=g["labels"].to(dev))
        out.loss.backward()
        torch.cuda.synchronize()
        if impl == name:
            assert hf_plugin.calls["interval"] + hf_plugin.calls["lds"] - before >= 4  # 2 encoder + 2 decoder layers went through libafk.so
        params = dict(m.named_parameters())
        res[impl] = dict(loss=abs(float(out.loss.detach()) - float(g["loss"])),
                         logit=float((out.logits.float().cpu()[sel] - g["logits_bf16"].float()).abs().max()),
                         grads={k: _rel(params[k].grad, v) for k, v in g["grads"].items()})
    floor, got = res["sdpa"], res[name]
    dist = _floor_distribution(dev, case)   # the stock-attention reference model in bf16 on this device, over the floor batches
    fl_loss = [floor["loss"]] + [abs(f["loss"] - o["loss_ref"]) for f, o in zip(dist["floor"], dist["ours"])]
    fl_logit = [floor["logit"]] + [f["logits"]["max"] for f in dist["floor"]]
    assert got["loss"] <= floor_bar(LOSS_ATOL, fl_loss), (got, fl_loss)
    assert got["logit"] <= floor_bar(logit_tol(g["logits_absmax"]), fl_logit), (got["logit"], fl_logit)
    bad = {}
    for k, v in got["grads"].items():
        fl = [floor["grads"][k]] + [f["grad_rel_l2"][k] for f in dist["floor"]]
        if min(fl) > NOISE_DOMINATED:
            continue   # bf16 cannot resolve this tensor on the sharp goldens (tests/_tol.py); pinned by the smooth cases
        # no cap here: both runs are the SAME bf16 reference model on the same batch and differ only in the attention kernels; where the stock run
        # is itself 1.26 rel-L2 off the fp32 golden (case C, conv1.weight) the plugin run is 1.26 off too - the bar is the stock run's range
        if v > floor_bar(GRAD_REL_L2, fl):
            bad[k] = (v, max(fl))
    assert not bad, bad


WIDE = dict(  # head_dim 64 (encoder) and 128 (decoder, GQA 4:2): the LDS-staged kernels of the AF3-7B geometry
    audio_config=dict(num_mel_bins=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512, hidden_size=256, max_source_positions=1500),
    text_config=dict(vocab_size=1024, hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                     max_position_embeddings=4096),
    audio_token_id=1023,
)


def _batch(dev, padded):
    torch.manual_seed(11)
    feats = (torch.randn(2, 128, 3000) * 0.5).to(torch.bfloat16)
    fmask = torch.ones(2, 3000, dtype=torch.int32)
    n1 = 750
    if padded:
        fmask[1, 2000:] = 0
        n1 = 500
    S = 6 + 750 + 10
    ids = torch.randint(0, 1000, (2, S))
    att = torch.ones(2, S, dtype=torch.int64)
    ids[0, 3:753] = 1023
    off = 750 - n1
    ids[1, 3 + off:753] = 1023
    if padded:  # left padding, as the processor produces
        ids[1, :off] = 0
        ids[1, off:off + 3] = torch.randint(0, 1000, (3,))
        att[1, :off] = 0
    labels = torch.full((2, S), -100)
    labels[:, -10:] = ids[:, -10:]
    return dict(input_ids=ids.to(dev), input_features=feats.to(dev), input_features_mask=fmask.to(dev),
                attention_mask=att.to(dev) if padded else None, labels=labels.to(dev)), att.bool()


@pytest.mark.parametrize("padded", [False, True])
def test_afk_attention_vs_sdpa_on_device(dev, padded):
    """unpadded -> mask None -> afk_attn2_* (full encoder / causal GQA decoder); padded -> bool masks the plugin recognises as "full / causal
    AND key in [lo_b, hi_b)" -> the same kernels with kv_lo / kv_len (masks of any other interval shape, other head sizes and Q != K take
    the interval kernels: test_generate_through_plugin, the tiny64 golden cases)"""
    from audio_flamingo_amd import hf_plugin

    name = hf_plugin.register()
    m = _hf_model(WIDE, dev, seed=3)
    m.train()
    batch, keep = _batch(dev, padded)
    res = {}
    before = dict(hf_plugin.calls)
    for impl in ("sdpa", name):
        m.set_attn_implementation(impl)
        m.zero_grad()
        out = m(**batch)
        out.loss.backward()
        torch.cuda.synchronize()
        res[impl] = (float(out.loss.detach()), out.logits.float().cpu()[keep], {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    served = {k: hf_plugin.calls[k] - before[k] for k in before}
    assert served == {"lds": 4, "interval": 0}, served
    assert abs(res["sdpa"][0] - res[name][0]) <= 1e-2, (res["sdpa"][0], res[name][0])
    assert float((res["sdpa"][1] - res[name][1]).abs().max()) <= 4e-2
    # k_proj.bias is left out: its true gradient is identically zero (a bias on k adds q.b to every score of a query - softmax does not see a
    # per-query constant), so both runs hold pure rounding noise there and their relative distance means nothing (0.061 against the 0.06 bar once
    # the kernels' summation order changed in round 4)
    bad = {k: _rel(res[name][2][k], v) for k, v in res["sdpa"][2].items()
           if v.float().norm() > 0 and not k.endswith("k_proj.bias") and _rel(res[name][2][k], v) > 6e-2}
    assert not bad, bad


def test_generate_through_plugin(dev):
    """greedy decode with the reference's KV cache: prefill (Q == K, causal) and Q = 1 steps (interval kernels over the cache)"""
    from audio_flamingo_amd import hf_plugin

    name = hf_plugin.register()
    m = _hf_model(WIDE, dev, seed=3).eval()
    batch, _ = _batch(dev, False)
    batch.pop("labels")
    ids = {}
    for impl in ("sdpa", name):
        m.set_attn_implementation(impl)
        with torch.no_grad():
            ids[impl] = m.generate(**batch, max_new_tokens=6, do_sample=False).cpu()
    # random-init logits have small top-1/top-2 gaps, and one differing token changes everything after it: require the first
    # generated token to match on every row and >= 90 % of all positions to agree
    n = batch["input_ids"].shape[1]
    assert ids["sdpa"].shape == ids[name].shape
    assert torch.equal(ids["sdpa"][:, :n + 1], ids[name][:, :n + 1]), (ids["sdpa"][:, n:], ids[name][:, n:])
    assert (ids["sdpa"] == ids[name]).float().mean() > 0.9, (ids["sdpa"][:, n:], ids[name][:, n:])
    assert hf_plugin.calls["interval"] > 0  # the Q = 1 decode steps
