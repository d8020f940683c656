"""CPU: pins the oracle restatement (oracle/af3_oracle.py) to the golden vectors generated from the live reference
implementation (oracle/make_golden.py -> tests/golden/), and to the live implementation when importable."""
import os

import numpy as np
import pytest
import torch

from oracle import af3_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CFG = dict(enc_heads=4, heads=4, kv_heads=2, eps=1e-6, theta=10000.0, audio_token_id=1023)


def _state():
    return {k: v.float() for k, v in torch.load(os.path.join(G, "tiny64_state_bf16.pt")).items()}


@pytest.mark.parametrize("case", ["A", "B"])
def test_oracle_matches_reference_golden(case):
    g = torch.load(os.path.join(G, f"tiny64_case{case}.pt"))
    sd = _state()
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    with torch.no_grad():
        out = O.forward(sd, CFG, g["ids"], g["feats"].float(), g["fmask"].long(), labels=g["labels"], attention_mask=g["att"])
    assert abs(float(out["loss"]) - float(g["loss"])) < 2e-5, (float(out["loss"]), float(g["loss"]))
    sel = g["labels"] != -100
    ref = g["logits_bf16"].float()
    got = out["logits"][sel]
    assert (got - ref).abs().max() < 2e-2 * ref.abs().max().clamp_min(1.0)  # golden logits are stored bf16-rounded
    keep = g["att"].bool()
    confident = (g["top_gap"] > 1e-3) & keep
    assert torch.equal(out["logits"].argmax(-1)[confident], g["argmax"][confident])
    a_ref = g["audio_bf16"].float()
    assert out["audio"].shape == a_ref.shape
    assert (out["audio"] - a_ref).abs().max() < 1e-2 * a_ref.abs().max()


def test_oracle_generate_matches_reference():
    g = torch.load(os.path.join(G, "tiny64_caseA.pt"))
    sd = _state()
    with torch.no_grad():
        ids = O.greedy_generate(sd, CFG, g["ids"][:1], g["feats"][:1].float(), g["fmask"][:1].long(), 2)
    assert ids[0, -2:].tolist() == g["generate"][0, -4:-2].tolist()


def test_oracle_logmel_matches_reference():
    g = torch.load(os.path.join(G, "logmel_case.pt"))
    w = torch.zeros(1, g["n"])
    w[0, : g["wave_head"].numel()] = g["wave_head"]
    f = O.log_mel(w)
    assert f.shape == (1, 128, 3000)
    assert (f[0, :, :: g["stride"]] - g["feats_sub"]).abs().max() < 1e-5


def test_mel_bank_matches_third_party():
    tf = pytest.importorskip("transformers")
    from transformers.audio_utils import mel_filter_bank

    ref = mel_filter_bank(num_frequency_bins=201, num_mel_filters=128, min_frequency=0.0, max_frequency=8000.0, sampling_rate=16000,
                          norm="slaney", mel_scale="slaney")
    assert np.abs(O.mel_filter_bank(128) - ref).max() < 1e-12
    from audio_flamingo_amd.frontend import mel_filter_bank as mine

    assert np.abs(mine(128) - ref).max() < 1e-12
