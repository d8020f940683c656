"""Generates tests/golden/tiny64_music_*.pt from the LIVE reference implementation of Music Flamingo (transformers 5.15
MusicFlamingoForConditionalGeneration, fp32 CPU).   Run in the build container:  python oracle/make_golden_music.py

One case: sample 0 = two windows (30 s + 10 s -> 750 + 250 <sound> tokens: window index 0 and 1 of the same sample), sample 1 = one 5 s
window (125 tokens), right-padded text; forward + backward; the rotary time embedding sees two window offsets and three lengths.
ORACLE tooling - test infrastructure only.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    from transformers import MusicFlamingoConfig, MusicFlamingoForConditionalGeneration

    from oracle.make_golden import TINY

    torch.manual_seed(20)
    T = {k: (dict(v, model_type="audioflamingo3_encoder") if k == "audio_config" else v) for k, v in TINY.items()}
    model = MusicFlamingoForConditionalGeneration(MusicFlamingoConfig(**T)).eval()
    g = torch.Generator().manual_seed(21)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(p.to(torch.bfloat16).float())
            if n.endswith(".bias"):
                p.copy_((0.02 * torch.randn(p.shape, generator=g)).to(torch.bfloat16).float())
            elif "norm" in n and n.endswith(".weight"):
                p.copy_((1 + 0.05 * torch.randn(p.shape, generator=g)).to(torch.bfloat16).float())
    torch.save({k: v.to(torch.bfloat16) for k, v in model.state_dict().items()}, os.path.join(OUT, "tiny64_music_state_bf16.pt"))
    feats = (torch.randn(3, 128, 3000, generator=g) * 0.5).to(torch.bfloat16)
    fmask = torch.ones(3, 3000, dtype=torch.int32)
    fmask[1, 1000:] = 0
    fmask[2, 500:] = 0
    S = 9 + 1000 + 9 + 12
    ids = torch.randint(0, 1000, (2, S), generator=g)
    att = torch.ones(2, S, dtype=torch.long)
    ids[0, 9:1009] = 1023
    ids[1, 9:134] = 1023
    ids[1, 155:] = 0
    att[1, 155:] = 0
    labels = torch.full((2, S), -100)
    labels[0, -12:] = ids[0, -12:]
    labels[1, 143:155] = ids[1, 143:155]
    out = model(input_ids=ids, input_features=feats.float(), input_features_mask=fmask.long(), attention_mask=att, labels=labels)
    out.loss.backward()
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    pick = ["lm_head.weight", "model.multi_modal_projector.linear_1.weight", "model.audio_tower.layer_norm.weight",
            "model.audio_tower.layers.1.fc1.weight", "model.audio_tower.layers.0.self_attn.q_proj.bias", "model.audio_tower.conv2.bias",
            "model.language_model.layers.0.self_attn.k_proj.weight"]
    with torch.no_grad():
        audio = model.get_audio_features(feats.float(), fmask.long(), input_ids=ids).pooler_output
    keep = labels != -100
    top2 = out.logits.detach().topk(2, -1).values
    torch.save(dict(feats=feats, fmask=fmask, ids=ids, att=att, labels=labels, loss=out.loss.detach(),
                    logits_bf16=out.logits.detach()[keep].to(torch.bfloat16), argmax=out.logits.detach().argmax(-1),
                    top_gap=top2[..., 0] - top2[..., 1], audio_bf16=audio.to(torch.bfloat16),
                    grads={k: grads[k].to(torch.bfloat16) for k in pick}, grad_norms={k: float(v.norm()) for k, v in grads.items()}),
               os.path.join(OUT, "tiny64_music_case.pt"))
    print("music loss", float(out.loss), "audio rows", audio.shape)


if __name__ == "__main__":
    sys.exit(main())
