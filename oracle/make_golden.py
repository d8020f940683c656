"""Generates tests/golden/*.pt from the LIVE reference implementation (transformers 5.15 AudioFlamingo3, fp32 CPU).

Run in the build container:  python oracle/make_golden.py        (single-threaded on purpose: bit-reproducible, ~5 min)
The outputs are small seeded input/output vectors that travel with the repo (the tests never need to re-run
the reference).  Config "tiny64": full architecture, reduced width/depth, dims that satisfy the MFMA kernels
(multiples of 64, head_dim 32/64).

Round 2: a random-init tiny model has nearly flat logits (|x| <= 1.2, top-1/top-2 gap below bf16 noise at 94 % of the
positions, greedy decoding emits one constant id), so "token indices bit-exact" was barely exercised.  The reference
model is therefore TRAINED here for a few hundred AdamW steps (the reference's own forward/backward, torch.optim) on a
synthetic language with a sharp answer at every position: text tokens follow a fixed random permutation chain
(next = PERM[cur]) and a <sound> placeholder is followed by another placeholder.  The trained weights are rounded to
bf16 and every golden below is computed from the rounded weights.

Cases: (A) full windows, no padding; (B) one padded window (5 s clip) + RIGHT-padded row: key-padding mask, token-count
formula, valid-row scatter; (C) a batch built by the reference's own AudioFlamingo3Processor (synthetic word-level
tokenizer, 5 s + 30 s clips) - LEFT padded, labels from output_labels=True.
"""
import math
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
AUDIO_ID, PAD_ID, N_WORDS = 1023, 1000, 256     # text tokens 0..255 ("w0".."w255"), <pad> = 1000, <sound> = 1023
TRAIN_STEPS, N_GEN = 400, 24


def perm_table():
    return torch.randperm(N_WORDS, generator=torch.Generator().manual_seed(123))


def chain(start, n, perm):
    out = [int(start)]
    for _ in range(n - 1):
        out.append(int(perm[out[-1]]))
    return torch.tensor(out, dtype=torch.long)


def round_bf16_(model):
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.to(torch.bfloat16).float())


def build(seed=0):
    from transformers import AudioFlamingo3Config, AudioFlamingo3ForConditionalGeneration

    torch.manual_seed(seed)
    cfg = AudioFlamingo3Config(**TINY)
    model = AudioFlamingo3ForConditionalGeneration(cfg).eval()
    with torch.no_grad():
        # give biases / norm weights non-trivial values (the default init leaves them 0 / 1)
        g = torch.Generator().manual_seed(seed + 1)
        for n, p in model.named_parameters():
            if n.endswith(".bias"):
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
            elif "norm" in n and n.endswith(".weight"):
                p.copy_(1 + 0.05 * torch.randn(p.shape, generator=g))
    return cfg, model


def train_sharp(model, steps=TRAIN_STEPS, log=True):
    """fine-tune the REFERENCE model with the reference's own forward / autograd / torch.optim.AdamW until its next-token
    distribution is sharp (the synthetic language above); cosine learning-rate decay 4e-3 -> 4e-4"""
    perm = perm_table()
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=4e-3, weight_decay=0.0)
    for step in range(steps):
        for gr in opt.param_groups:
            gr["lr"] = 4e-4 + 0.5 * (4e-3 - 4e-4) * (1 + math.cos(math.pi * step / steps))
        gg = torch.Generator().manual_seed(1000 + step)
        B, nt = 2, 125
        feats = torch.randn(B, 128, 3000, generator=gg) * 0.5
        fmask = torch.zeros(B, 3000, dtype=torch.long)
        fmask[:, :500] = 1                                  # 5 s clips: 125 <sound> tokens
        rows = []
        for b in range(B):
            c = chain(int(torch.randint(0, N_WORDS, (1,), generator=gg)), 126, perm)
            cut = int(torch.randint(3, 12, (1,), generator=gg))
            rows.append(torch.cat([c[:cut], torch.full((nt,), AUDIO_ID), c[cut:]]))
        ids = torch.stack(rows)
        out = model(input_ids=ids, input_features=feats, input_features_mask=fmask, labels=ids.clone())
        opt.zero_grad()
        out.loss.backward()
        opt.step()
        if log and step % 50 == 0:
            print(f"  train step {step}: loss {float(out.loss.detach()):.4f}", flush=True)
    model.eval()
    return model


def make_inputs(case, seed=1234):
    """seed 1234 = the stored goldens A / B; other seeds = the batches of the multi-seed noise-floor measurement
    (tests/test_model_gpu.py::_floor_distribution)"""
    from transformers import WhisperFeatureExtractor

    rng = np.random.default_rng(seed)
    perm = perm_table()
    fe = WhisperFeatureExtractor(feature_size=128)
    if case == "A":  # 2 samples x one full 30 s window
        waves = [rng.standard_normal(480000).astype(np.float32) * 0.1 for _ in range(2)]
        n_tok = [750, 750]
    else:  # one full window + one 5 s clip (500 valid frames -> 125 tokens)
        waves = [rng.standard_normal(480000).astype(np.float32) * 0.1, rng.standard_normal(80000).astype(np.float32) * 0.1]
        n_tok = [750, 125]
    out = fe(waves, sampling_rate=16000, return_attention_mask=True, padding="max_length", return_tensors="pt")
    feats, fmask = out["input_features"], out["attention_mask"]
    S = 9 + 750 + 9 + 24
    ids = torch.zeros((2, S), dtype=torch.long)
    att = torch.ones((2, S), dtype=torch.long)
    labels = torch.full((2, S), -100, dtype=torch.long)
    for i in range(2):
        c = chain(int(rng.integers(0, N_WORDS)), 42, perm)
        row = torch.cat([c[:9], torch.full((n_tok[i],), AUDIO_ID), c[9:]])
        ids[i, : len(row)] = row
        att[i, len(row):] = 0  # right padding (case B second row)
        labels[i, len(row) - 24: len(row)] = row[-24:]
        if len(row) < S:
            ids[i, len(row):] = 0
    return dict(feats=feats, fmask=fmask, ids=ids, att=att, labels=labels)


def make_processor():
    """the reference's own processor with a synthetic offline tokenizer"""
    from tokenizers import Tokenizer
    from tokenizers.models import WordLevel
    from tokenizers.pre_tokenizers import WhitespaceSplit
    from transformers import AudioFlamingo3Processor, PreTrainedTokenizerFast, WhisperFeatureExtractor

    vocab = {f"w{i}": i for i in range(1000)}
    vocab.update({"<pad>": PAD_ID, "<unk>": 1001, "<sound>": AUDIO_ID})
    vocab.update({f"<r{i}>": i for i in range(1002, AUDIO_ID)})  # no holes in the id space
    tok = Tokenizer(WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = WhitespaceSplit()
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, pad_token="<pad>", unk_token="<unk>", additional_special_tokens=["<sound>"])
    return AudioFlamingo3Processor(WhisperFeatureExtractor(feature_size=128), fast)


def make_inputs_processor(seed=4321):
    """case C: what AudioFlamingo3Processor.__call__ hands to the model for two (prompt, clip) pairs of different lengths:
    left-padded input_ids / attention_mask, labels with -100 on <sound> and <pad> positions (seed 4321 = the stored golden)"""
    rng = np.random.default_rng(seed)
    perm = perm_table()
    proc = make_processor()
    waves = [rng.standard_normal(80000).astype(np.float32) * 0.1, rng.standard_normal(480000).astype(np.float32) * 0.1]
    texts = []
    for i, n in enumerate((20, 31)):
        c = chain(int(rng.integers(0, N_WORDS)), n, perm).tolist()
        texts.append(" ".join(f"w{t}" for t in c[:5]) + " <sound> " + " ".join(f"w{t}" for t in c[5:]))
    out = proc(text=texts, audio=waves, output_labels=True)
    return dict(feats=out["input_features"], fmask=out["input_features_mask"], ids=out["input_ids"], att=out["attention_mask"],
                labels=out["labels"])


PICK = ["lm_head.weight", "model.language_model.embed_tokens.weight", "model.language_model.layers.0.self_attn.k_proj.weight",
        "model.language_model.layers.1.mlp.up_proj.weight", "model.language_model.layers.0.input_layernorm.weight",
        "model.multi_modal_projector.linear_1.weight", "model.audio_tower.layers.0.self_attn.q_proj.bias",
        "model.audio_tower.layers.1.fc1.weight", "model.audio_tower.conv1.weight", "model.audio_tower.conv2.bias",
        "model.audio_tower.layer_norm.weight"]


def golden_case(model, inp, generate_from=None):
    # the stored feats are bf16-rounded: the reference outputs are computed on exactly those inputs
    fe_b = inp["feats"].to(torch.bfloat16).float()
    model.zero_grad()
    out = model(input_ids=inp["ids"], input_features=fe_b, input_features_mask=inp["fmask"], attention_mask=inp["att"], labels=inp["labels"])
    out.loss.backward()
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    gen = None
    with torch.no_grad():
        audio = model.get_audio_features(fe_b, inp["fmask"]).pooler_output
        if generate_from is not None:
            gen = model.generate(input_ids=generate_from["ids"], input_features=fe_b[: generate_from["n_windows"]],
                                 input_features_mask=inp["fmask"][: generate_from["n_windows"]], attention_mask=generate_from["att"],
                                 max_new_tokens=N_GEN, do_sample=False)
    keep = inp["labels"] != -100  # logits are stored only where the loss looks at them (argmax / top-2 gap are stored everywhere)
    logits = out.logits.detach()
    top2 = logits.topk(2, -1).values
    return dict(
        feats=inp["feats"].to(torch.bfloat16), fmask=inp["fmask"].to(torch.int32), ids=inp["ids"], att=inp["att"], labels=inp["labels"],
        loss=out.loss.detach(), logits_bf16=logits[keep].to(torch.bfloat16), argmax=logits.argmax(-1), top_gap=(top2[..., 0] - top2[..., 1]),
        logits_absmax=float(logits.abs().max()), audio_bf16=audio.to(torch.bfloat16),
        grads={k: grads[k].to(torch.bfloat16) for k in PICK}, grad_norms={k: float(v.norm()) for k, v in grads.items()}, generate=gen)


SMOOTH_SEED = 77


def make_inputs_smooth(case):
    """cases D / E: the batches of A / B (own seed) with a label on EVERY text position - 2 x 41 next-token targets instead of 2 x 24 - so that
    the loss gradient is an average over many positions of a flat (random-init) distribution"""
    inp = make_inputs("A" if case == "D" else "B", seed=SMOOTH_SEED + (0 if case == "D" else 1))
    ids, att = inp["ids"], inp["att"]
    labels = torch.where((ids != AUDIO_ID) & att.bool(), ids, torch.full_like(ids, -100))
    inp["labels"] = labels
    return inp


def golden_case_smooth(model, inp):
    """forward + backward of the live reference (fp32 CPU): loss, logits on the label rows, audio rows and the gradient of EVERY parameter"""
    fe_b = inp["feats"].to(torch.bfloat16).float()
    model.zero_grad()
    out = model(input_ids=inp["ids"], input_features=fe_b, input_features_mask=inp["fmask"], attention_mask=inp["att"], labels=inp["labels"])
    out.loss.backward()
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    with torch.no_grad():
        audio = model.get_audio_features(fe_b, inp["fmask"]).pooler_output
    sh = torch.nn.functional.pad(inp["labels"], (0, 1), value=-100)[:, 1:]
    keep = sh != -100           # the rows the loss reads (position i predicts token i + 1)
    logits = out.logits.detach()
    # conditioning record: the reference's OWN bf16 run (eager PyTorch CPU) against the fp32 gradients above, per tensor
    import copy

    mb = copy.deepcopy(model).to(torch.bfloat16)
    mb.zero_grad()
    mb(input_ids=inp["ids"], input_features=fe_b.to(torch.bfloat16), input_features_mask=inp["fmask"], attention_mask=inp["att"],
       labels=inp["labels"]).loss.backward()
    pb = dict(mb.named_parameters())
    ref_bf16 = {k: float((pb[k].grad.float() - v).norm() / v.norm().clamp_min(1e-20)) for k, v in grads.items()}
    return dict(ref_bf16_cpu_rel=ref_bf16,feats=inp["feats"].to(torch.bfloat16), fmask=inp["fmask"].to(torch.int32), ids=inp["ids"], att=inp["att"], labels=inp["labels"],
                loss=out.loss.detach(), logits_bf16=logits[keep].to(torch.bfloat16), logits_absmax=float(logits.abs().max()),
                audio_bf16=audio.to(torch.bfloat16), grads={k: v.to(torch.bfloat16) for k, v in grads.items()},
                grad_norms={k: float(v.norm()) for k, v in grads.items()})


def main_smooth(out_dir=OUT):
    """Round 3: SMOOTH goldens beside the sharp ones.  Same architecture (TINY, 2 + 2 layers), RANDOM-INIT weights
    (reference _init_weights N(0, 0.02) + non-trivial biases / norm weights, rounded to bf16): logits O(1), loss ~ ln(vocab) - the loss surface
    is smooth, bf16 rounding moves a gradient by ~1 %, so EVERY parameter gradient is held to the fixed bar of tests/_tol.py (6e-2 rel-L2,
    no noise-floor relaxation).  The sharp goldens A/B/C test token ids and logits where they mean something; these test the backward."""
    torch.set_num_threads(1)
    os.makedirs(out_dir, exist_ok=True)
    cfg, model = build(seed=SMOOTH_SEED)
    with torch.no_grad():
        # With N(0, 0.02) projections the ENCODER's attention logits have a standard deviation of 0.05: every softmax row is uniform to three
        # digits and the q / k gradients of its last layer vanish to the size of bf16 rounding noise (|g| 4e-4 against 1e-2 elsewhere; the
        # reference's own bf16 run is 18 % off on them, measured).  A golden for the BACKWARD needs attention that attends: the encoder's q and k
        # projections are scaled by 4 (logit std ~ 0.8), everything else stays as initialised.
        for n, p in model.named_parameters():
            if "audio_tower" in n and (n.endswith("q_proj.weight") or n.endswith("k_proj.weight")):
                p.mul_(4.0)
    round_bf16_(model)
    torch.save({k: v.to(torch.bfloat16) for k, v in model.state_dict().items()}, os.path.join(out_dir, "tiny64_smooth_state_bf16.pt"))
    model.train()  # dropout 0 everywhere: train() == eval() numerically; the backward is what is stored
    for case in ("D", "E"):
        gold = golden_case_smooth(model, make_inputs_smooth(case))
        torch.save(gold, os.path.join(out_dir, f"tiny64_case{case}.pt"))
        print(case, "loss", float(gold["loss"]), "logits |max|", round(gold["logits_absmax"], 3), "label rows", int(gold["logits_bf16"].shape[0]),
              "gradient tensors", len(gold["grads"]), flush=True)


def main(out_dir=OUT):
    torch.set_num_threads(1)  # bit-reproducible goldens (reduction order of the CPU GEMMs depends on the thread count)
    os.makedirs(out_dir, exist_ok=True)
    cfg, model = build()
    train_sharp(model)
    round_bf16_(model)  # the GPU path stores bf16: both sides share identical (rounded) parameters
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    torch.save({k: v.to(torch.bfloat16) for k, v in sd.items()}, os.path.join(out_dir, "tiny64_state_bf16.pt"))
    for case in ("A", "B", "C"):
        inp = make_inputs(case) if case != "C" else make_inputs_processor()
        gen_from = None
        if case == "A":    # sample 0, prompt + audio + the first 4 answer tokens -> greedy continuation along the chain
            gen_from = dict(ids=inp["ids"][:1, : 9 + 750 + 9 + 4], att=inp["att"][:1, : 9 + 750 + 9 + 4], n_windows=1)
        elif case == "C":  # the processor's left-padded batch as it is (both rows)
            gen_from = dict(ids=inp["ids"], att=inp["att"], n_windows=inp["feats"].shape[0])
        gold = golden_case(model, inp, gen_from)
        torch.save(gold, os.path.join(out_dir, f"tiny64_case{case}.pt"))
        valid = inp["att"].bool()
        conf = (gold["top_gap"] > 0.25) & valid
        print(case, "loss", float(gold["loss"]), "logits |max|", round(gold["logits_absmax"], 2), "confident",
              int(conf.sum()), "/", int(valid.sum()), "gen tail", None if gold["generate"] is None else gold["generate"][0, -N_GEN:].tolist())
    # log-mel golden: 10 s of signal + silence -> features from the reference feature extractor
    from transformers import WhisperFeatureExtractor

    rng = np.random.default_rng(7)
    w = (rng.standard_normal(480000) * 0.1).astype(np.float32)
    w[160000:] = 0.0  # 10 s of signal, rest silence (exercises the max-8 floor)
    fe = WhisperFeatureExtractor(feature_size=128)
    f = fe._torch_extract_fbank_features(w[None])
    torch.save(dict(wave_head=torch.from_numpy(w[:160000].copy()), feats_sub=torch.from_numpy(f[0, :, ::37].copy()), stride=37,
                    n=480000), os.path.join(out_dir, "logmel_case.pt"))
    for fn in sorted(os.listdir(out_dir)):
        print(fn, os.path.getsize(os.path.join(out_dir, fn)) // 1024, "KiB")


if __name__ == "__main__":
    # python oracle/make_golden.py [out_dir]          -> sharp goldens A/B/C (+ log-mel), ~4 min
    # python oracle/make_golden.py smooth [out_dir]   -> smooth goldens D/E (random-init, every parameter gradient), ~20 s
    if len(sys.argv) > 1 and sys.argv[1] == "smooth":
        sys.exit(main_smooth(*sys.argv[2:3]))
    sys.exit(main(*sys.argv[1:2]))
