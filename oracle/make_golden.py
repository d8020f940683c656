"""Generates tests/golden/*.pt from the LIVE reference implementation (transformers 5.15 AudioFlamingo3, fp32 CPU).

Run in the build container:  python oracle/make_golden.py
The outputs are small seeded input/output vectors that travel with the repo (the tests never need to re-run
the reference).  Config "tiny64": full architecture, reduced width/depth, dims that satisfy the MFMA kernels
(multiples of 64, head_dim 32/64).  Two cases: (A) full windows, no padding; (B) one padded window (5 s clip)
exercising the key-padding mask, the token-count formula and the valid-row scatter.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")

TINY = dict(
    audio_config=dict(num_mel_bins=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256, hidden_size=128,
                      max_source_positions=1500),
    text_config=dict(vocab_size=1024, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                     num_key_value_heads=2, max_position_embeddings=4096),
    audio_token_id=1023,
)


def build(seed=0):
    from transformers import AudioFlamingo3Config, AudioFlamingo3ForConditionalGeneration

    torch.manual_seed(seed)
    cfg = AudioFlamingo3Config(**TINY)
    model = AudioFlamingo3ForConditionalGeneration(cfg).eval()
    # weights are rounded to bf16 so that the GPU path (bf16 storage) and the fp32 references share identical parameters
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.to(torch.bfloat16).float())
        # give biases / norm weights non-trivial values (the default init leaves them 0 / 1)
        g = torch.Generator().manual_seed(seed + 1)
        for n, p in model.named_parameters():
            if n.endswith(".bias"):
                p.copy_((0.02 * torch.randn(p.shape, generator=g)).to(torch.bfloat16).float())
            elif "norm" in n and n.endswith(".weight"):
                p.copy_((1 + 0.05 * torch.randn(p.shape, generator=g)).to(torch.bfloat16).float())
    return cfg, model


def make_inputs(case):
    from transformers import WhisperFeatureExtractor

    rng = np.random.default_rng(1234)
    fe = WhisperFeatureExtractor(feature_size=128)
    if case == "A":  # 2 samples x one full 30 s window
        waves = [rng.standard_normal(480000).astype(np.float32) * 0.1 for _ in range(2)]
        n_tok = [750, 750]
    else:  # one full window + one 5 s clip (500 valid frames -> 125 tokens)
        waves = [rng.standard_normal(480000).astype(np.float32) * 0.1, rng.standard_normal(80000).astype(np.float32) * 0.1]
        n_tok = [750, 125]
    out = fe(waves, sampling_rate=16000, return_attention_mask=True, padding="max_length", return_tensors="pt")
    feats, fmask = out["input_features"], out["attention_mask"]
    wav_pad = np.zeros((2, 480000), np.float32)
    for i, w in enumerate(waves):
        wav_pad[i, : len(w)] = w
    S = 9 + 750 + 9 + 24
    ids = torch.zeros((2, S), dtype=torch.long)
    att = torch.ones((2, S), dtype=torch.long)
    labels = torch.full((2, S), -100, dtype=torch.long)
    for i in range(2):
        text = lambda n: torch.from_numpy(rng.integers(0, 1000, n))
        row = torch.cat([text(9), torch.full((n_tok[i],), 1023), text(9), text(24)])
        ids[i, : len(row)] = row
        att[i, len(row):] = 0  # right padding (case B second row)
        labels[i, len(row) - 24: len(row)] = row[-24:]
        if len(row) < S:
            ids[i, len(row):] = 0
    return dict(wave=torch.from_numpy(wav_pad), feats=feats, fmask=fmask, ids=ids, att=att, labels=labels)


def main():
    os.makedirs(OUT, exist_ok=True)
    cfg, model = build()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    torch.save({k: v.to(torch.bfloat16) for k, v in sd.items()}, os.path.join(OUT, "tiny64_state_bf16.pt"))
    for case in ("A", "B"):
        inp = make_inputs(case)
        kw = dict(input_ids=inp["ids"], input_features=inp["feats"], input_features_mask=inp["fmask"], attention_mask=inp["att"])
        model.zero_grad()
        out = model(**kw, labels=inp["labels"])
        out.loss.backward()
        grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        pick = ["lm_head.weight", "model.language_model.layers.0.self_attn.k_proj.weight", "model.language_model.layers.1.mlp.up_proj.weight",
                "model.language_model.layers.0.input_layernorm.weight", "model.multi_modal_projector.linear_1.weight",
                "model.audio_tower.layers.0.self_attn.q_proj.bias", "model.audio_tower.layers.1.fc1.weight", "model.audio_tower.conv1.weight",
                "model.audio_tower.conv2.bias", "model.audio_tower.layer_norm.weight"]
        with torch.no_grad():
            gen = model.generate(**{k: v[:1] for k, v in kw.items()}, max_new_tokens=4, do_sample=False) if case == "A" else None
            audio = model.get_audio_features(inp["feats"], inp["fmask"]).pooler_output
        keep = inp["labels"] != -100  # logits are stored only where the loss looks at them (argmax is stored everywhere)
        gold = dict(
            wave=inp["wave"][:, :160000].clone() if case == "A" else inp["wave"].clone()[:, :80000],  # enough to re-derive nothing: feats are stored
            feats=inp["feats"].to(torch.bfloat16), fmask=inp["fmask"].to(torch.int32), ids=inp["ids"], att=inp["att"], labels=inp["labels"],
            loss=out.loss.detach(), logits_bf16=out.logits.detach()[keep].to(torch.bfloat16),
            argmax=out.logits.detach().argmax(-1), audio_bf16=audio.to(torch.bfloat16),
            grads={k: grads[k].to(torch.bfloat16) for k in pick}, grad_norms={k: float(v.norm()) for k, v in grads.items()},
            generate=gen,
        )
        # the stored feats are bf16-rounded: recompute the reference outputs on exactly those inputs
        fe_b = inp["feats"].to(torch.bfloat16).float()
        model.zero_grad()
        out = model(input_ids=inp["ids"], input_features=fe_b, input_features_mask=inp["fmask"], attention_mask=inp["att"], labels=inp["labels"])
        out.loss.backward()
        grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        with torch.no_grad():
            audio = model.get_audio_features(fe_b, inp["fmask"]).pooler_output
            gen = model.generate(input_ids=inp["ids"][:1], input_features=fe_b[:1], input_features_mask=inp["fmask"][:1],
                                 attention_mask=inp["att"][:1], max_new_tokens=4, do_sample=False) if case == "A" else None
        top2 = out.logits.detach().topk(2, -1).values
        gold.update(loss=out.loss.detach(), logits_bf16=out.logits.detach()[keep].to(torch.bfloat16), argmax=out.logits.detach().argmax(-1),
                    top_gap=(top2[..., 0] - top2[..., 1]), audio_bf16=audio.to(torch.bfloat16),
                    grads={k: grads[k].to(torch.bfloat16) for k in pick}, grad_norms={k: float(v.norm()) for k, v in grads.items()},
                    generate=gen)
        del gold["wave"]
        torch.save(gold, os.path.join(OUT, f"tiny64_case{case}.pt"))
        print(case, "loss", float(out.loss), "gen", None if gen is None else gen[0, -4:].tolist())
    # log-mel golden: 2 s of the case-A waveform -> features from the reference feature extractor
    from transformers import WhisperFeatureExtractor

    rng = np.random.default_rng(7)
    w = (rng.standard_normal(480000) * 0.1).astype(np.float32)
    w[160000:] = 0.0  # 10 s of signal, rest silence (exercises the max-8 floor)
    fe = WhisperFeatureExtractor(feature_size=128)
    f = fe._torch_extract_fbank_features(w[None])
    torch.save(dict(wave_head=torch.from_numpy(w[:160000].copy()), feats_sub=torch.from_numpy(f[0, :, ::37].copy()), stride=37,
                    n=480000), os.path.join(OUT, "logmel_case.pt"))
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)) // 1024, "KiB")


if __name__ == "__main__":
    sys.exit(main())
