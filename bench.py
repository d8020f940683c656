#!/usr/bin/env python
"""bench.py - AF3-7B bf16 training throughput on MI355X (BASELINE.json metric: audio-seconds/s + decoder tokens/s).

    python bench.py --gpus 1 --steps K --warmup W                    (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W                   (N>1, one rank per GPU, RCCL over xGMI)

One "step" = one full training step of the hot path on one micro-batch of synthetic input already resident in HBM:
waveform -> log-mel kernel -> AF-Whisper encoder -> pool+LN -> projector -> <sound> scatter -> Qwen2.5-7B decoder ->
fused lm_head+CE -> full backward (every tower trainable, stage-3 fine-tune) -> [DP: bucketed gradient all-reduce
overlapped with backward] -> fused AdamW.  Workload = BASELINE.json configs[1]/[2]: per sample one 30 s
clip (480 000 samples, 0.1*N(0,1), seed 1234+idx) and S=1024 tokens (9 prompt + 750 <sound> + 9 prompt + 256 answer, loss on
the answer), micro-batch 8 per GPU, random-init weights N(0, 0.02) of the AF3-7B architecture.

Rank 0 prints ONE JSON line.  value = audio-seconds/s over all ranks; decoder tokens/s are reported beside it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The step runs on three HIP streams (compute, wgrad, optimizer/all-reduce) and RCCL adds its own: with ROCm's default of 4 hardware
# queues per process two of them can end up sharing a queue and serialise (measured: 1-rank run with the RCCL process group alive
# 467 ms/step -> 447 with 8 queues).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

S_TOK, N_AUDIO_TOK, N_ANSWER, AUDIO_ID = 1024, 750, 256, 151669
CLIP_SECONDS = 30.0
# BASELINE.json configs[1]/[2] ("clip30": one 30 s window per sample, micro-batch 8) and configs[4] ("long5min": one 5-minute clip =
# 10 full windows per sample, 7 500 <sound> tokens, S = 7 774, micro-batch 1, per-layer activation checkpointing on both towers).
WORKLOADS = {"clip30": dict(windows=1, batch=8, checkpoint=False), "long5min": dict(windows=10, batch=1, checkpoint=True),
             # AF3's stated maximum ("up to 10 minutes", /root/reference README.md:109): 20 windows, 15 000 <sound> rows, S = 15 274
             "long10min": dict(windows=20, batch=1, checkpoint=True),
             # BASELINE configs[3]: AF1/AF2-style ICL step (Perceiver resampler + gated cross-attention, 4 clips per sample); shapes are
             # builder-declared (audio_flamingo_amd/flamingo_icl.py ICL4), parity w.r.t. AF1/AF2 UNPINNED
             "icl4": dict(windows=0, batch=8, checkpoint=False)}
ICL_S, ICL_ANSWER, ICL_CLIP_SECONDS = 512, 128, 10.0


def af3_7b_config(enc_layers=32, dec_layers=28):
    from transformers import AudioFlamingo3Config

    return AudioFlamingo3Config(
        audio_config=dict(num_mel_bins=128, num_hidden_layers=enc_layers, num_attention_heads=20, intermediate_size=5120, hidden_size=1280,
                          max_source_positions=1500),
        text_config=dict(vocab_size=152064, hidden_size=3584, intermediate_size=18944, num_hidden_layers=dec_layers, num_attention_heads=28,
                         num_key_value_heads=4, max_position_embeddings=32768, rms_norm_eps=1e-6,
                         rope_parameters=dict(rope_theta=1000000.0, rope_type="default")),
        audio_token_id=AUDIO_ID,
    )


def synthetic_batch(batch, first_idx, device, windows=1):
    """SURVEY.md §8(d) synthetic inputs; sample i uses numpy default_rng(1234 + i).  A sample is `windows` consecutive full 30 s windows
    (waves [batch * windows, 480 000], window-major inside a sample) and 9 prompt + 750*windows <sound> + 9 prompt + 256 answer ids."""
    n_audio = N_AUDIO_TOK * windows
    S = 9 + n_audio + 9 + N_ANSWER
    waves = np.empty((batch * windows, int(CLIP_SECONDS * 16000)), np.float32)
    ids = np.empty((batch, S), np.int64)
    for b in range(batch):
        rng = np.random.default_rng(1234 + first_idx + b)
        waves[b * windows:(b + 1) * windows] = (0.1 * rng.standard_normal((windows, waves.shape[1]))).astype(np.float32)
        text = rng.integers(0, 151643, size=9 + 9 + N_ANSWER)
        ids[b] = np.concatenate([text[:9], np.full(n_audio, AUDIO_ID), text[9:18], text[18:]])
    labels = ids.copy()
    labels[:, : S - N_ANSWER] = -100
    return (torch.from_numpy(waves).to(device), torch.from_numpy(ids).to(device), torch.from_numpy(labels).to(device))


def train_flops_per_sample(S=S_TOK, windows=1, recompute=False):
    """algorithmic FLOPs (2*MACs, causal attention at 1/2), forward x 3; recompute adds the checkpointed layers' second
    forward (x 4 on those layers) - reported separately as hardware FLOPs, never as model FLOPs.  recompute: False | True (every layer) |
    (n_enc, n_dec) = how many layers of each tower the memory-budgeted plan recomputes"""
    enc_1 = 2 * 1500 * (4 * 1280 ** 2 + 2 * 1280 * 5120) + 4 * 1500 ** 2 * 1280
    enc = windows * (2 * 3000 * 128 * 3 * 1280 + 2 * 1500 * 1280 * 3 * 1280)
    proj = windows * 2 * 750 * (1280 * 3584 + 3584 * 3584)
    dec_1 = 2 * S * (2 * 3584 ** 2 + 2 * 3584 * 512 + 3 * 3584 * 18944) + 2 * S ** 2 * 3584
    lm = 2 * S * 3584 * 152064
    n_enc, n_dec = (32, 28) if recompute is True else (0, 0) if not recompute else recompute
    # the re-run of a checkpointed layer does not compute the GEMM whose only product is its (discarded) output - encoder fc2, decoder down_proj
    # (functional.RECOMPUTE_SKIP_TAIL, round 6): the second forward of those layers is counted without it
    from audio_flamingo_amd import functional as _F
    skip = bool(getattr(_F, "RECOMPUTE_SKIP_TAIL", False))
    enc_rerun = enc_1 - (2 * 1500 * 1280 * 5120 if skip else 0)
    dec_rerun = dec_1 - (2 * S * 3584 * 18944 if skip else 0)
    return 3.0 * (enc + proj + lm) + windows * (enc_1 * 3.0 * 32 + enc_rerun * n_enc) + dec_1 * 3.0 * 28 + dec_rerun * n_dec


def _layer_flops():
    """forward FLOPs of one encoder layer (per 30 s window) and one decoder layer (S = 1024): the weights of the depth extrapolation"""
    fe = 2 * 1500 * (4 * 1280 ** 2 + 2 * 1280 * 5120) + 4 * 1500 ** 2 * 1280
    fd = 2 * S_TOK * (2 * 3584 ** 2 + 2 * 3584 * 512 + 3 * 3584 * 18944) + 2 * S_TOK ** 2 * 3584
    return fe, fd


def _pick_cpu_threads(cap):
    """eager CPU kernels do not scale to every hardware thread of a 2-socket host (r01: 256 threads ran the 1+1-layer probe 5x slower than
    8 threads of the build container): take the thread count that runs a gate/up-sized fp32 GEMM fastest"""
    a, b = torch.randn(1024, 3584), torch.randn(18944, 3584)
    best, best_t = 1, float("inf")
    for n in (8, 16, 32, 64, 128, 256):
        if n > cap:
            break
        torch.set_num_threads(n)
        torch.nn.functional.linear(a, b)
        t0 = time.perf_counter()
        for _ in range(3):
            torch.nn.functional.linear(a, b)
        t = time.perf_counter() - t0
        if t < 0.92 * best_t:
            best, best_t = n, t
    return best


def cpu_baseline(budget_s=25.0):
    """The REFERENCE implementation itself (transformers.AudioFlamingo3ForConditionalGeneration, SURVEY.md §8d protocol) timed on this
    host's cores on a bounded sample of the same workload: full WIDTH, depth-reduced (2 encoder + 2 decoder layers, then 1 + 1), one
    sample (one 30 s window, S = 1024: 9 prompt + 750 <sound> + 9 prompt + 256 answer, loss on the answer), forward + backward, eager
    PyTorch CPU, in fp32 and in bf16, 1 warm-up + up to 3 timed iterations each (bounded by budget_s per dtype).  The full-depth figure is
    EXTRAPOLATED linearly in layer count: T(32, 28) = T(1,1) + (T(2,2) - T(1,1)) * (31 fE + 27 fD) / (fE + fD), fE / fD = FLOPs of an
    encoder / decoder layer (the two depths separate the per-layer cost from stem + projector + lm_head + loss).  No optimizer step.
    A reported baseline, not a target."""
    from transformers import AudioFlamingo3ForConditionalGeneration

    ncpu = os.cpu_count() or 1
    threads = _pick_cpu_threads(ncpu)
    torch.set_num_threads(threads)
    fe, fd = _layer_flops()
    ids = torch.randint(0, 151643, (1, S_TOK))
    ids[0, 9: 9 + N_AUDIO_TOK] = AUDIO_ID
    labels = ids.clone()
    labels[:, : S_TOK - N_ANSWER] = -100
    feats = torch.randn(1, 128, 3000) * 0.5
    times, detail = {}, {}
    for depth in (2, 1):
        with torch.device("meta"):
            model = AudioFlamingo3ForConditionalGeneration(af3_7b_config(depth, depth))
        model = model.to_empty(device="cpu")
        with torch.no_grad():
            for n_, p_ in model.named_parameters():
                if n_.endswith("norm.weight") or n_.endswith("layer_norm.weight"):
                    p_.fill_(1.0)
                elif n_.endswith(".bias"):
                    p_.zero_()
                else:
                    p_.normal_(0.0, 0.02)
        model.train()
        for dt in (torch.float32, torch.bfloat16):
            model = model.to(dt)
            kw = dict(input_ids=ids, input_features=feats.to(dt), input_features_mask=torch.ones(1, 3000, dtype=torch.long), labels=labels)
            ts, t_start = [], time.perf_counter()
            for it in range(4):  # 1 warm-up + up to 3 timed
                t0 = time.perf_counter()
                model.zero_grad(set_to_none=True)
                model(**kw).loss.backward()
                dtm = time.perf_counter() - t0
                if it > 0:
                    ts.append(dtm)
                if it >= 1 and time.perf_counter() - t_start > budget_s * (1.0 if depth == 2 else 0.5):
                    break
            key = "fp32" if dt == torch.float32 else "bf16"
            times[(depth, key)] = sum(ts) / len(ts)
            detail[f"T({depth},{depth})_{key}_s"] = [round(t, 3) for t in ts]
        del model
    scale = (31 * fe + 27 * fd) / (fe + fd)
    full = {k: times[(1, k)] + max(times[(2, k)] - times[(1, k)], 0.0) * scale for k in ("fp32", "bf16")}
    cfg1 = None
    try:
        cfg1 = config1_generate_cpu()
    except Exception as e:
        cfg1 = {"error": repr(e)[:200]}
    return {
        "config1_generate": cfg1,
        "value": CLIP_SECONDS / full["bf16"], "unit": "audio-s/s", "cores": threads, "kind": "reference",
        "decoder_tokens_per_s": S_TOK / full["bf16"], "dtype": "bf16",
        "fp32": {"value": CLIP_SECONDS / full["fp32"], "decoder_tokens_per_s": S_TOK / full["fp32"], "s_per_sample_extrapolated": round(full["fp32"], 2)},
        "s_per_sample_extrapolated": round(full["bf16"], 2), "host_cpus": ncpu, "timings": detail,
        "sample": (f"transformers AudioFlamingo3ForConditionalGeneration, eager PyTorch CPU ({threads} threads of {ncpu} hardware threads), full width, "
                   f"B=1 (one 30 s window, S=1024), fwd+bwd, 1 warm-up + {len(detail['T(2,2)_bf16_s'])} timed at 2+2 layers and at 1+1 layers; "
                   f"bf16: T(2,2)={times[(2, 'bf16')]:.2f}s T(1,1)={times[(1, 'bf16')]:.2f}s, fp32: T(2,2)={times[(2, 'fp32')]:.2f}s "
                   f"T(1,1)={times[(1, 'fp32')]:.2f}s; EXTRAPOLATED linearly in layer count (FLOP-weighted enc/dec split) to 32+28 layers: "
                   f"{full['bf16']:.1f}s (bf16) / {full['fp32']:.1f}s (fp32) per sample, no optimizer step"),
    }


def config1_generate_cpu():
    """SURVEY.md §8(d) CPU protocol step (1) = BASELINE configs[0]: the reference's end-to-end generate() on the host - 1 x 5 s synthetic wav, batch 1,
    PyTorch CPU eager, the tiny preset of SURVEY Appendix B-1 (plumbing: feature extractor -> encoder -> projector -> <sound> scatter -> decoder with
    KV cache -> greedy ids).  1 warm-up + 3 timed calls of feature extraction + generate(max_new_tokens=16)."""
    from transformers import AudioFlamingo3Config, AudioFlamingo3ForConditionalGeneration, WhisperFeatureExtractor

    torch.manual_seed(0)
    cfg = AudioFlamingo3Config(audio_config=dict(num_mel_bins=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, hidden_size=64,
                                                 max_source_positions=1500),
                               text_config=dict(vocab_size=1000, hidden_size=96, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                                                num_key_value_heads=2, max_position_embeddings=4096), audio_token_id=999)
    model = AudioFlamingo3ForConditionalGeneration(cfg).eval()
    wav = np.random.default_rng(0).standard_normal(80000).astype(np.float32) * 0.1
    fe = WhisperFeatureExtractor(feature_size=128)
    ids = torch.tensor([[1, 2, 3] + [999] * 125 + [4, 5, 6, 7]])
    ts, new = [], 16
    for it in range(4):
        t0 = time.perf_counter()
        f = fe(wav, sampling_rate=16000, return_attention_mask=True, padding="max_length", return_tensors="pt")
        with torch.no_grad():
            out = model.generate(input_ids=ids, input_features=f["input_features"], input_features_mask=f["attention_mask"], max_new_tokens=new, do_sample=False)
        if it > 0:
            ts.append(time.perf_counter() - t0)
    t = sum(ts) / len(ts)
    return {"what": "BASELINE configs[0]: reference AF3 generate(), 1 x 5 s synthetic wav, batch 1, PyTorch CPU eager, tiny preset (SURVEY Appendix B-1): plumbing only",
            "seconds_per_call": t, "new_tokens": int(out.shape[1] - ids.shape[1]), "audio_s_per_s": 5.0 / t, "new_tokens_per_s": new / t, "timed_calls": len(ts)}


def eager_rocm_baseline(dev, feats, ids, labels, steps=3, clip=0.0):
    """The same-node "before" number (SURVEY.md §8d last row, BASELINE.md §2 "B-rocm-eager"): the UNMODIFIED reference model
    (transformers.AudioFlamingo3ForConditionalGeneration, attn_implementation sdpa, rocBLAS/hipBLASLt GEMMs, MIOpen convs) at full depth
    in bf16 on this MI355X with torch.optim.AdamW (fused), on the same synthetic batch (micro-batch halved until it fits).  1 warm-up +
    `steps` timed steps.  Also: the vendor GEMM (what F.linear dispatches to) on the gate|up shape 8192 x 37888 x 3584, HIP-event timed."""
    from transformers import AudioFlamingo3ForConditionalGeneration

    torch.manual_seed(0)
    with torch.device(dev):
        model = AudioFlamingo3ForConditionalGeneration(af3_7b_config()).to(torch.bfloat16)
    from tools.parity_fulldepth import restore_rope_buffers

    restore_rope_buffers(model)   # .to(bf16) rounds the rotary inv_freq buffer; a from_pretrained(dtype=bf16) model keeps it in fp32
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, fused=True)
    B = ids.shape[0]
    res = None
    while B >= 1 and res is None:
        kw = dict(input_ids=ids[:B], input_features=feats[:B], input_features_mask=torch.ones(B, feats.shape[-1], device=dev, dtype=torch.long),
                  labels=labels[:B])
        try:
            ts = []
            for it in range(steps + 2):   # two untimed steps (allocator growth, the vendor libraries' first-use heuristics), then `steps` timed ones
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                opt.zero_grad(set_to_none=True)
                loss = model(**kw).loss
                loss.backward()
                if clip > 0:
                    torch.nn.utils.clip_grad_norm_(model.parameters(), clip)  # the reference recipe's clip (HF Trainer max_grad_norm)
                opt.step()
                torch.cuda.synchronize()
                if it > 1:
                    ts.append(time.perf_counter() - t0)
            # the MEDIAN step: one run of the round measured this leg at 1 595 ms and another at 768 (a step that hit an allocator stall) beside the usual 634-645 -
            # a mean would have turned that into a 4.2 x "speedup"; the median is the conservative figure for the reference
            ms = 1000.0 * sorted(ts)[len(ts) // 2]
            res = {"ms_per_step": ms, "step_ms": [round(1000.0 * t, 1) for t in ts], "micro_batch": B, "value": B * CLIP_SECONDS / (ms * 1e-3), "unit": "audio-s/s",
                   "decoder_tokens_per_s": B * ids.shape[1] / (ms * 1e-3), "loss_last": float(loss.detach()),
                   "model_tflops_per_gpu": train_flops_per_sample(ids.shape[1]) * B / (ms * 1e-3) / 1e12,
                   "peak_mem_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1), "max_grad_norm": clip if clip > 0 else None}
        except torch.OutOfMemoryError:
            opt.zero_grad(set_to_none=True)
            torch.cuda.empty_cache()
            B //= 2
    del model, opt
    torch.cuda.empty_cache()
    # vendor GEMM on the largest shape of the step (decoder gate|up projection)
    a = torch.randn(8192, 3584, device=dev, dtype=torch.bfloat16)
    w = torch.randn(37888, 3584, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        torch.nn.functional.linear(a, w)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        torch.nn.functional.linear(a, w)
    e1.record()
    torch.cuda.synchronize()
    gemm_ms = e0.elapsed_time(e1) / 10
    # our kernel on the SAME operands in the same process (operand statistics move the power-limited clock: compare like with like)
    from audio_flamingo_amd import ops as _ops

    c = torch.empty((8192, 37888), device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        _ops.gemm_nt(a, w, out=c)
    e0.record()
    for _ in range(10):
        _ops.gemm_nt(a, w, out=c)
    e1.record()
    torch.cuda.synchronize()
    afk_ms = e0.elapsed_time(e1) / 10
    out = res or {"ms_per_step": None, "value": None, "unit": "audio-s/s", "micro_batch": 0}
    out.update({"kind": "reference model, eager PyTorch-ROCm (sdpa, vendor BLAS), torch.optim.AdamW(fused), bf16, full depth 32+28",
                "vendor_gemm_gate_up": {"shape": [8192, 37888, 3584], "ms": gemm_ms, "tflops": 2.0 * 8192 * 37888 * 3584 / (gemm_ms * 1e-3) / 1e12,
                                        "what": "torch.nn.functional.linear (rocBLAS / hipBLASLt) on N(0,1) bf16 operands, HIP events, 10 launches",
                                        "afk_same_operands_ms": afk_ms, "afk_same_operands_tflops": 2.0 * 8192 * 37888 * 3584 / (afk_ms * 1e-3) / 1e12},
                "steps": steps, "warmup": 1})
    return out


def run_icl4(args, dev):
    """BASELINE configs[3] as a measured workload: one training step (fwd + bwd + fused AdamW) of the AF1/AF2-style in-context model of
    audio_flamingo_amd/flamingo_icl.py on micro-batch `args.batch` x 4 clips; synthetic encoder features N(0,1) [B, 4, 64, 768] (the
    audio encoder itself - AF-CLAP - is out of scope: SURVEY.md §2.2), S = 512 text tokens with 4 <audio> markers, loss on the last 128."""
    from audio_flamingo_amd import ops
    from audio_flamingo_amd.flamingo_icl import ICL4, FlamingoICLForCausalLM, TensorAdamW

    c = dict(ICL4)
    if args.dec_layers != 28:
        c["layers"] = args.dec_layers
    B = args.batch
    model = FlamingoICLForCausalLM(c, device=dev, seed=0)
    opt = TensorAdamW(model.parameters(), lr=1e-5)
    g = torch.Generator(device=dev).manual_seed(1234)
    feats = torch.randn((B, c["clips"], c["enc_frames"], c["enc_dim"]), device=dev, generator=g).to(torch.bfloat16)
    ids = torch.randint(0, 151643, (B, ICL_S), device=dev, generator=g)
    ids[:, [8, 136, 264, 392]] = c["audio_marker_id"]
    labels = ids.clone()
    labels[:, : ICL_S - ICL_ANSWER] = -100
    shift = torch.nn.functional.pad(labels, (0, 1), value=-100)[:, 1:].reshape(-1)
    rows = (shift != -100).nonzero().reshape(-1)

    def step():
        opt.zero_grad()
        loss = model(ids, feats, labels, label_rows=rows)
        loss.backward()
        opt.step()
        return loss

    first = float(step().detach())
    for _ in range(max(args.warmup - 1, 0)):
        step()
    torch.cuda.synchronize()
    ops.prof_reset()
    ops.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.prof_enable(False)
    gemm_ms, gemm_flops, gemm_launches = ops.prof_collect()
    sps = B * args.steps / dt
    nparams = sum(p.numel() for p in model.parameters())
    ach = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    return {
        "metric": "audio-sec/s + decoder tokens/s, AF1/AF2-style ICL bf16 train (BASELINE configs[3], builder-declared shapes, parity UNPINNED)",
        "value": sps * c["clips"] * ICL_CLIP_SECONDS, "unit": "audio-s/s", "decoder_tokens_per_s": sps * ICL_S, "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": (f"ICL step: {c['clips']} clips/sample ({ICL_CLIP_SECONDS:.0f} s each, encoder features [{c['enc_frames']}, {c['enc_dim']}]) -> Perceiver "
                                f"resampler ({c['n_latents']} latents, depth {c['resampler_depth']}) -> {c['layers']}-layer Qwen2.5-3B-class decoder with a "
                                f"gated cross-attention block every {c['xattn_every']} layers, S={ICL_S}, fwd+bwd+AdamW"),
                   "micro_batch_per_gpu": B, "global_batch": B, "seq_len": ICL_S, "parallelism": "dp1", "params": nparams, "shapes": c},
        "loss": float(loss.detach()), "loss_first_step": first, "host_enqueue_ms_per_step": round(1000.0 * host / args.steps, 1),
        "gemm_executed_tflops_per_step": gemm_flops / args.steps / 1e12, "peak_mem_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
        "roofline": {"bound": "mfma", "kernel": "gemm_nt_bf16_k256 / gemm_xt_bf16_k256 (+ k128 / split-K for small shapes)", "achieved": ach, "peak": 2500.0,
                     "unit": "TFLOP/s", "frac": ach / 2500.0, "traffic": None, "launches": gemm_launches, "gemm_ms_per_step": gemm_ms / args.steps,
                     "note": "HIP events around every GEMM launch of the timed steps (single stream)"},
    }


def decode_leg(model, frontend, waves, ids, dev):
    """generate() throughput on the resident AF3-7B replica (greedy, KV cache, HIP-graph-replayed decode step)"""
    feats = frontend(waves, out_dtype=torch.bfloat16)
    prompt = ids[:, : 9 + N_AUDIO_TOK + 9]
    weight_bytes = 2.0 * (28 * (3584 * 4608 + 3584 * 3584 + 3 * 3584 * 18944) + 152064 * 3584)   # decoder Linears + lm_head, bf16, read once per token
    out = {"prompt_tokens": int(prompt.shape[1]), "weight_bytes_per_token": weight_bytes, "hbm_roofline_ms_per_token": weight_bytes / 8e12 * 1e3}
    out["protocol"] = ("decode_ms_per_token = (t_33 - t_1) / 32 with t_n = one generate(max_new_tokens=n) call, as in rounds 2-3: it carries the eager first step and the one-time "
                       "graph capture of every call; decode_ms_per_token_steady = (t_97 - t_33) / 64: replayed steps only (keys 801 -> 865)")
    reps = (32 + prompt.shape[0] - 1) // prompt.shape[0]   # B = 32 (round 6: four groups of eight sequences per launch): the batch's prompts repeated
    prompt32, feats32 = prompt.repeat(reps, 1), feats.repeat(reps, *([1] * (feats.dim() - 1)))
    for B in (1, 8, 32):
        t = {}
        for new in (1, 33, 97):
            model.generate(prompt32[:B], input_features=feats32[:B], max_new_tokens=new)  # warm (allocator, one-time library init)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.generate(prompt32[:B], input_features=feats32[:B], max_new_tokens=new)
            torch.cuda.synchronize()
            t[new] = time.perf_counter() - t0
        ms = 1e3 * (t[33] - t[1]) / 32
        steady = 1e3 * (t[97] - t[33]) / 64
        out[f"B{B}"] = {"prefill_plus_first_token_ms": 1e3 * t[1], "decode_ms_per_token": ms, "decode_tokens_per_s": B / (ms * 1e-3),
                        "frac_of_weight_streaming_roofline": out["hbm_roofline_ms_per_token"] / ms,
                        "decode_ms_per_token_steady": steady, "decode_tokens_per_s_steady": B / (steady * 1e-3),
                        "frac_of_weight_streaming_roofline_steady": out["hbm_roofline_ms_per_token"] / steady}
    return out


LINE_LIMIT = 4096   # bytes: the driver parses the LAST stdout line; round 5's 30 KB line came back unparsed


def _r(x, n=4):
    return round(x, n) if isinstance(x, float) else x


def _pick(d, keys, n=4):
    return None if not isinstance(d, dict) else {k: _r(d[k], n) for k in keys if k in d}


def parity_brief(s):
    """<= 600 bytes of the full-depth parity summary (tools/parity_fulldepth.summary): verdict + the headline distances"""
    if not isinstance(s, dict):
        return None
    if "error" in s:
        return {"green": None, "error": str(s["error"])[:160]}
    lg, g = s.get("logits", {}), s.get("gradients", {})
    out = {"green": s.get("green"), "loss_abs_err": _r(s.get("loss_abs_err"), 5), "logits_rel_l2": _r(lg.get("rel_l2"), 4),
           "logits_rel_l2_floor": _r(s.get("logits_floor_ref_bf16", {}).get("rel_l2"), 4),
           "argmax_mismatches_on_confident_rows": lg.get("argmax_mismatches_on_confident_rows"), "confident_rows": lg.get("confident_rows"),
           "rows": lg.get("rows"), "grad_tensors": g.get("tensors"),
           "grad_ratio_median": _r(g.get("ours_over_floor", {}).get("median"), 3), "grad_ratio_max": _r(g.get("ours_over_floor", {}).get("max"), 3),
           "bucket_norm_rel_err_max": _r(g.get("bucket_norm_rel_err_max"), 5),
           "failed_checks": [k for k, v in (s.get("checks") or {}).items() if not v]}
    for k in ("peaked", "long5min_train"):      # the round-6 legs: each already a short dict
        if k in s:
            out[k] = s[k]
    return out


def compact_line(res, detail_path=None):
    """The ONE JSON line the driver parses: the contract's fields + roofline + roofline_hbm + cpu_baseline + parity verdict, <= LINE_LIMIT bytes.
    Everything else of `res` (long-audio / decode / ICL / eager legs, DP prose, per-tensor tables) lives in gpurun_out/bench_detail.json."""
    c = res.get("config") or {}
    rf, rh, cb = res.get("roofline"), res.get("roofline_hbm"), res.get("cpu_baseline")
    out = {k: _r(res.get(k), 3) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                          "dtype", "data")}
    out["config"] = {k: c[k] for k in ("workload", "micro_batch_per_gpu", "global_batch", "seq_len", "parallelism", "activation_checkpointing") if k in c}
    for k in ("decoder_tokens_per_s", "answer_tokens_per_s", "model_tflops_per_gpu", "executed_tflops_per_gpu", "hardware_tflops_per_gpu",
              "model_frac_of_mfma_peak", "executed_frac_of_mfma_peak", "speedup_vs_eager_rocm", "peak_mem_gib"):
        if res.get(k) is not None:
            out[k] = _r(res[k], 4)
    out["loss"], out["loss_first_step"], out["step_enqueue"] = _r(res.get("loss"), 4), _r(res.get("loss_first_step"), 4), res.get("step_enqueue")
    if rf:
        out["roofline"] = _pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_ms", "gemm_ms_per_step",
                                     "frac_of_power_ceiling", "power_ceiling_tflops"))
        out["roofline"]["kernel"] = str(out["roofline"].get("kernel"))[:120]
    if rh:
        out["roofline_hbm"] = _pick(rh, ("bound", "kernel", "achieved", "peak", "unit", "frac", "ms"))
        out["roofline_hbm"]["kernel"] = str(out["roofline_hbm"].get("kernel"))[:60]
    if cb:
        out["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "dtype", "decoder_tokens_per_s", "s_per_sample_extrapolated", "host_cpus"), 5)
        out["cpu_baseline"]["sample"] = str(cb.get("sample"))[:420]
    eb = res.get("eager_rocm_baseline")
    if isinstance(eb, dict) and eb.get("value"):
        out["eager_rocm_baseline"] = _pick(eb, ("value", "unit", "ms_per_step", "micro_batch"), 2)
    if res.get("parity_fulldepth") is not None:
        out["parity_fulldepth"] = parity_brief(res["parity_fulldepth"])
    la = res.get("long_audio_configs4")
    if isinstance(la, dict):
        out["long_audio_configs4"] = _pick(la, ("ms_per_step", "value", "unit", "decoder_tokens_per_s"), 2)
    dec = res.get("decode")
    if isinstance(dec, dict) and "B1" in dec:
        out["decode_ms_per_token_steady"] = {b: _r(dec[b]["decode_ms_per_token_steady"], 3) for b in ("B1", "B8", "B32") if b in dec}
    dp = res.get("dp")
    if isinstance(dp, dict):
        out["dp"] = {k: _r(dp[k], 2) for k in ("backend", "comm", "form", "chosen", "probe_ms", "exposed_comm_ms", "buckets", "launched_by") if k in dp}
        if isinstance(dp.get("preflight"), dict):
            out["dp"]["preflight"] = {k: _pick(v, ("ok", "failed", "allreduce_ms", "allreduce_busbw_gbs", "reduce_scatter_ms", "all_gather_ms", "skipped"), 2)
                                      for k, v in dp["preflight"].items()}
        if dp.get("fallback"):
            out["dp"]["fallback"] = [str(x)[:120] for x in dp["fallback"]][:3]
        out["replicas_identical_after_steps"] = res.get("replicas_identical_after_steps")
    for k in ("param_checksum", "rank_losses", "rccl_ranks", "stream_priorities"):    # tests/test_dp_gpu.py compares schedules on these
        if res.get(k) is not None:
            out[k] = res[k]
    if res.get("dry_run"):
        for k in ("dry_run", "replicas_identical_after_steps", "collective_backend", "buckets", "collectives_per_step", "launched_by"):
            if k in res:
                out[k] = res[k]
    if detail_path:
        out["detail"] = detail_path
    line = json.dumps(out)
    if len(line) >= LINE_LIMIT:   # never emit an unparseable record: shed the optional parts, largest first
        for k in ("parity_fulldepth", "eager_rocm_baseline", "long_audio_configs4", "decode_ms_per_token_steady", "dp"):
            if k in out and len(line) >= LINE_LIMIT:
                out[k] = {"moved_to": detail_path} if k != "parity_fulldepth" else {"green": out[k].get("green"), "moved_to": detail_path}
                line = json.dumps(out)
    assert len(line) < LINE_LIMIT and json.loads(line)["metric"] == res.get("metric"), f"bench line is {len(line)} bytes"
    return line


def write_detail(res, name="bench_detail.json"):
    """the full record beside the line: gpurun_out/ (merged back by gpurun); best effort - returns the repo-relative path or None"""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name), "w") as f:
            json.dump(res, f, indent=1)
        return "gpurun_out/" + name
    except OSError:
        return None


def mfma_power_ceiling(dev, seconds=1.2):
    """what the matrix pipe sustains on THIS chip under its power cap when nothing but MFMAs run (csrc/ceiling.hip mode 0: v_mfma_f32_32x32x16_bf16 on
    N(0,1) operands resident in registers, 256 workgroups x 8 waves, no LDS / global access in the loop): 0.3 s ramp + `seconds` timed with HIP events.
    The datasheet 2.5 PFLOP/s assumes 2.4 GHz; at the 1.4 kW cap this loop runs at ~1.83 GHz (profiles/r06_mfma_power_ceiling.json)."""
    import ctypes

    from audio_flamingo_amd import _lib

    g = torch.Generator(device=dev).manual_seed(0)
    src = torch.randn(1 << 20, device=dev, generator=g).to(torch.bfloat16)
    sink = torch.zeros(4, device=dev)
    fl = ctypes.c_double(0)
    st = torch.cuda.current_stream().cuda_stream

    def go():
        _lib.call("afk_mfma_ceiling", 0, 256, 20000, src.data_ptr(), sink.data_ptr(), ctypes.byref(fl), st)

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        go()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n, t0 = 0, time.perf_counter()
    e0.record()
    while time.perf_counter() - t0 < seconds:
        go()
        n += 1
        if n % 8 == 0:
            torch.cuda.current_stream().synchronize()
    e1.record()
    torch.cuda.synchronize()
    return fl.value * n / (e0.elapsed_time(e1) * 1e-3) / 1e12


def dry_run_cpu(args):
    """CONTROL-FLOW TEST ONLY - no GPU, no kernel, no throughput (value = null).  Runs the N > 1 sequence of main() over gloo on the host:
    process group -> replica = the real model class on the CPU (layout / arena / buckets only) -> DataParallelEngine + parameter broadcast ->
    K steps whose "backward" is a stub that writes rank-dependent gradients and reports the blocks in the order the real backward does ->
    bucket all-reduces issued from the arena's ready callbacks -> a stub SGD update (torch, host) in place of the fused AdamW kernel ->
    replica checksum all-gather -> ONE JSON line from rank 0.  tests/test_host_cpu.py::test_bench_multi_rank_control_flow_gloo launches it with
    2 processes through torch.distributed.run exactly as the driver launches the real thing."""
    import torch.distributed as dist

    from audio_flamingo_amd.dp import DataParallelEngine
    from audio_flamingo_amd.modeling import AudioFlamingo3ForConditionalGeneration
    from transformers import AudioFlamingo3Config

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tiny = dict(audio_config=dict(num_mel_bins=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256, hidden_size=128, max_source_positions=1500),
                text_config=dict(vocab_size=1024, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                                 max_position_embeddings=4096), audio_token_id=1023)
    model = AudioFlamingo3ForConditionalGeneration(AudioFlamingo3Config(**tiny), device="cpu", init_seed=10 + rank)   # replicas differ until the broadcast
    arena = model.arena
    engine = DataParallelEngine(arena, overlap=True)
    engine.broadcast_parameters(0)
    pf = engine.preflight()     # the known-answer round of every collective, as the real N > 1 run does before its first step
    assert pf["ok"], pf
    t0 = time.perf_counter()
    for k in range(args.warmup + args.steps):
        arena.zero_grad()
        engine.begin_backward()
        g = torch.Generator().manual_seed(1000 * k + rank)
        for blk in reversed(arena.order):                      # the real backward completes blocks in reverse layout order
            blk.grad.copy_(torch.randn(blk.shape, generator=g).to(torch.bfloat16))
            arena.grad_written(blk)                             # -> on_bucket_ready -> all-reduce of the finished bucket
        engine.finish()
        assert sorted(engine.issued) == list(range(len(arena.bucket_names))), engine.issued
        arena.params.add_((arena.grads.float() * engine.grad_scale).to(torch.bfloat16), alpha=-1e-3)   # stub update on the averaged gradients
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    p32 = arena.params.float()
    chk = torch.stack([p32.sum().double(), p32.abs().sum().double()])
    allc = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(allc, chk)
    identical = all(bool(torch.equal(c, allc[0])) for c in allc)
    assert identical, f"replicas diverged: {[c.tolist() for c in allc]}"
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(compact_line({"metric": "audio-sec/s + decoder tokens/s, AF3-7B bf16 train", "dry_run": True, "value": None, "unit": "audio-s/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "bf16", "data": "synthetic", "config": {"workload": "CONTROL-FLOW DRY RUN on the host (gloo): no kernels ran, nothing was measured"},
                          "replicas_identical_after_steps": identical, "collective_backend": "gloo", "buckets": len(arena.bucket_names),
                          "collectives_per_step": len(engine.issued) + 1, "launched_by": os.environ.get("AFK_BENCH_LAUNCHED_BY", "external"),
                          "dp": {"backend": "gloo", "preflight": {"torch": pf}}}), flush=True)


def self_spawn(n):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): re-exec this very command line under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` - one rank per GPU - and hand its
    exit status back.  The launcher form stays the primary one (it sets WORLD_SIZE, so this is never entered twice)."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, AFK_BENCH_LAUNCHED_BY="self_spawn", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "8")
    print(f"[bench] --gpus {n} without a launcher: re-executing under torch.distributed.run (port {port})", file=sys.stderr, flush=True)
    sys.exit(subprocess.call(cmd, env=env))


def settle_hbm(dev, quiet_s=8.0, timeout_s=60.0):
    """The amdgpu driver releases and scrubs a dead process's VRAM asynchronously (measured here: `mem_info_vram_used` falls from 215 GB
    to 0.3 GB over ~7 s after a 185 GB replica exits).  A replica that starts allocating inside or right behind that window gets
    fragmented small-page mappings and runs 30-60 % slower for its whole life (measured: 5 back-to-back bench processes -> runs 4, 5 at
    680-740 ms/step instead of 450; with 15 s between the processes all five at 457-460).  So before building the model, wait (untimed)
    until this GPU's VRAM has been empty and unchanged for `quiet_s` seconds."""
    import glob

    path = None
    try:
        bus = torch.cuda.get_device_properties(dev).pci_bus_id  # HIP: int; sysfs: 0000:BB:00.0
        for c in glob.glob("/sys/class/drm/card*/device"):
            real = os.path.realpath(c)
            if isinstance(bus, int) and real.split(":")[-2:-1] == [f"{bus:02x}"] and os.path.exists(c + "/mem_info_vram_used"):
                path = c + "/mem_info_vram_used"
    except Exception:
        path = None
    t0 = time.time()
    quiet_since, last = None, None
    while time.time() - t0 < timeout_s:
        if path is not None:
            used = int(open(path).read())
        else:
            free_b, total_b = torch.cuda.mem_get_info(dev)
            used = total_b - free_b
        now = time.time()
        if used < (2 << 30) and (last is None or abs(used - last) < (64 << 20)):
            quiet_since = quiet_since or now
            if now - quiet_since >= quiet_s:
                break
        else:
            quiet_since = None
        last = used
        time.sleep(0.5)
    return round(time.time() - t0, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="clip30", help="clip30 = BASELINE configs[1]/[2] (the headline); long5min = configs[4]; long10min = AF3's stated maximum clip length; icl4 = configs[3] (AF1/AF2-style ICL step)")
    ap.add_argument("--batch", type=int, default=0, help="micro-batch (samples) per GPU; default 8 for clip30 (BASELINE config), 1 for long5min")
    ap.add_argument("--no-checkpoint", action="store_true", help="long5min only: keep all activations instead of per-layer recompute")
    ap.add_argument("--enc-layers", type=int, default=32)
    ap.add_argument("--dec-layers", type=int, default=28)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eager-baseline", action="store_true", help="skip the unmodified reference model on this GPU (eager PyTorch-ROCm)")
    ap.add_argument("--eager-only", action="store_true", help="measure ONLY the unmodified reference model on this GPU (eager PyTorch-ROCm): for rocprofv3 kernel tables")
    ap.add_argument("--clip", type=float, default=0.0, help="global-norm gradient clipping (HF Trainer default: 1.0); 0 = off, as the eager reference leg runs")
    ap.add_argument("--no-graph", action="store_true", help="enqueue every step from Python (the default whenever stream priorities are on, i.e. unless AFK_STREAM_PRIORITIES=0)")
    ap.add_argument("--graph", action="store_true", help="N = 1: replay the captured HIP graph of the step instead of enqueueing it from Python.  Round 6 measured the eager "
                    "enqueue with stream priorities FASTER (387-397 ms against 400-405 for any graph variant, same boxes: profiles/r06_stream_priorities.md) - a graph "
                    "replay ends only when its lowest-priority branch has, and hipGraph maps the captured branches onto its own queues - so the graph is opt-in")
    ap.add_argument("--no-long-audio", action="store_true", help="skip the extra BASELINE configs[4] measurement (5-minute clips) and the 10-minute leg of the default run")
    ap.add_argument("--no-parity", action="store_true", help="skip the untimed full-depth parity leg of the default run (tools/parity_fulldepth.py: this model against the live "
                    "reference in fp32 and bf16 on the BASELINE configs[1] batch, one shared state_dict; forward-only on the configs[4] shape) -> `parity_fulldepth`")
    ap.add_argument("--parity-fulldepth", action="store_true", help="run ONLY the full-depth parity leg and print its summary as the JSON line")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the decode leg (KV-cache generate, B = 1 and 8) and the configs[3] ICL leg of the default run")
    ap.add_argument("--batches", type=int, default=0, help="distinct synthetic batches resident in HBM, one per step (0 = warm-up + steps + 8: no batch is ever "
                    "trained on twice inside the run, so nothing is memorised in the timed region; 1 = the same batch every step)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend for N > 1 / --force-dp: nccl = RCCL (production); "
                    "gloo = host-staged exchange, for running the N > 1 control flow with several ranks on ONE GPU (tests/test_dp_gpu.py)")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--no-opt-overlap", action="store_true", help="run AdamW after backward instead of bucket-by-bucket inside it")
    ap.add_argument("--no-wgrad-stream", action="store_true", help="keep the weight-gradient branch on the compute stream")
    ap.add_argument("--force-dp", action="store_true", help="run the data-parallel engine even with one rank AND issue its collectives (identity at world 1): the whole RCCL path on a 1-GPU box")
    ap.add_argument("--dp-graph", action="store_true", help="data parallel: capture the step - RCCL collectives included - into the HIP graph (default for N > 1: eager enqueue)")
    ap.add_argument("--cu-contention", default="", help="pre-flight of the multi-GPU run (VERDICT r03 item 5a): comma-separated CU counts, e.g. 0,8,16,32,64 - "
                    "for each, park that many persistent workgroups on a side stream (afk_cu_hog: what RCCL's channel kernels do to the GEMM rounds) and time "
                    "5 steps beside them; reported as cu_contention {n: ms_per_step}")
    ap.add_argument("--no-preflight", action="store_true", help="N > 1: skip the communicator pre-flight (known-answer all-reduce / reduce-scatter / all-gather on a layer bucket)")
    ap.add_argument("--no-dp-probe", action="store_true", help="N > 1: skip the startup probe that times the exchange forms (c10d all-reduce, native reduce-scatter + "
                    "all-gather, sharded optimizer) and run the first usable one; AFK_DP_COMM / AFK_DP_FORM set explicitly have the same effect")
    ap.add_argument("--no-settle", action="store_true", help="do not wait for the driver to release a previous process's VRAM (tests)")
    ap.add_argument("--detail-name", default="bench_detail.json", help="file under gpurun_out/ that receives the FULL record (every leg, every note); the "
                    "stdout line is the <= 4 KB summary of it")
    ap.add_argument("--dry-run-cpu", action="store_true", help="CONTROL-FLOW TEST ONLY (tests/test_host_cpu.py): no GPU, no kernels - the N > 1 sequence of this script "
                    "(process group, parameter broadcast, per-bucket exchange in backward order, replica checksum, one JSON line on rank 0) with a stub in place of the step")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_spawn(args.gpus)
    if args.dry_run_cpu:
        return dry_run_cpu(args)
    wl = WORKLOADS[args.workload]
    args.batch = args.batch or wl["batch"]
    windows, ckpt = wl["windows"], wl["checkpoint"] and not args.no_checkpoint
    n_audio_tok = N_AUDIO_TOK * windows
    s_tok = 9 + n_audio_tok + 9 + N_ANSWER

    import torch.distributed as dist

    from audio_flamingo_amd import ops
    from audio_flamingo_amd.arena import FusedAdamW
    from audio_flamingo_amd.dp import BackwardOverlap, DataParallelEngine
    from audio_flamingo_amd.frontend import LogMelFrontend
    from audio_flamingo_amd.modeling import AudioFlamingo3ForConditionalGeneration

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    local_rank %= max(torch.cuda.device_count(), 1)  # --backend gloo: several ranks may share one GPU (control-flow tests on 1-GPU boxes)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    waited = 0.0 if args.no_settle else settle_hbm(dev)
    if args.parity_fulldepth:
        from tools import parity_fulldepth as _pf

        rec = _pf.run(dev)
        _pf.write_record(rec)
        print(json.dumps({"parity_fulldepth": _pf.summary(rec)}), flush=True)
        return
    if args.eager_only:
        from audio_flamingo_amd.frontend import LogMelFrontend

        waves, ids, labels = synthetic_batch(args.batch, 0, dev, 1)
        feats_b = LogMelFrontend(dev)(waves, out_dtype=torch.bfloat16)
        print(json.dumps({"eager_rocm_baseline": eager_rocm_baseline(dev, feats_b, ids, labels, steps=args.steps, clip=args.clip)}), flush=True)
        return
    if args.workload == "icl4":
        assert world == 1, "icl4 is a single-GPU measurement"
        res = run_icl4(args, dev)
        res["waited_for_free_hbm_s"] = waited
        print(compact_line(res, write_detail(res, "bench_detail_icl4.json")), flush=True)
        return
    use_dp = world > 1 or args.force_dp
    if use_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(args.backend, **({"device_id": dev} if args.backend == "nccl" else {}))
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")   # where the script's own tiny collectives (timings, checksums) live

    full_model = args.enc_layers == 32 and args.dec_layers == 28
    model = AudioFlamingo3ForConditionalGeneration(af3_7b_config(args.enc_layers, args.dec_layers), device=dev, init_seed=0)
    model.check_placeholders = False  # the count assertion is a host sync; shapes are static in this benchmark
    if ckpt:   # --workload long5min: AFK_CKPT_POLICY picks the plan (default "full" = the reference's every-layer recompute; "budget" = memory-budgeted)
        model.gradient_checkpointing_enable()
    engine = opt = overlap = None
    opt_kw = dict(lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    dp_info = {"launched_by": os.environ.get("AFK_BENCH_LAUNCHED_BY", "external")}

    def configure_dp(kind, form):
        """(re)build the exchange engine + the optimizer that matches its form + the in-backward overlap driver.  kind: "torch" (c10d -> RCCL) | "native"
        (libafk.so's own communicator); form: "allreduce" | "rs_ag" (replicated AdamW) | "rs_adamw_ag" (optimizer sharded over the ranks)."""
        nonlocal engine, opt, overlap
        engine = opt = overlap = None
        import gc as _gc

        _gc.collect()
        torch.cuda.empty_cache()
        engine = DataParallelEngine(model.arena, overlap=not args.no_overlap, comm=kind, form=form)
        engine.force_collectives = engine.force_collectives or args.force_dp
        opt = engine.make_optimizer(**opt_kw)
        if args.clip > 0:
            opt.clip_norm = args.clip
        opt.sync_master()
        overlap = None if args.no_opt_overlap else BackwardOverlap(model.arena, opt, engine)
        if overlap is not None and os.environ.get("AFK_THIN_BLOCKS"):
            overlap.thin_blocks = int(os.environ["AFK_THIN_BLOCKS"])

    if use_dp:
        # N > 1: an explicit AFK_DP_COMM / AFK_DP_FORM is honoured as given; otherwise pre-flight the communicators (native first, the
        # c10d one as the fallback) and let a short startup probe pick the exchange form (below, once the step exists)
        env_kind, env_form = os.environ.get("AFK_DP_COMM"), os.environ.get("AFK_DP_FORM")
        staged = args.backend != "nccl"
        kinds = [env_kind] if env_kind else (["torch"] if staged else ["native", "torch"])
        dp_info["preflight"], dp_info["fallback"] = {}, []
        usable = []
        for kind in kinds:
            try:
                configure_dp(kind, env_form or ("rs_ag" if kind == "native" else "allreduce"))
                if not usable:
                    engine.broadcast_parameters(0)
                    opt.sync_master()
                pf = engine.preflight() if not args.no_preflight else {"ok": True, "skipped": True}
                dp_info["preflight"][kind] = pf
                if pf["ok"]:
                    usable.append(kind)
                else:
                    dp_info["fallback"].append(f"{kind}: pre-flight failed {pf['failed']}")
            except Exception as e:   # a communicator that cannot even be created / used: named, and the next one is tried
                dp_info["fallback"].append(f"{kind}: {type(e).__name__}: {str(e)[:160]}")
        if not usable:
            raise RuntimeError(f"no usable data-parallel communicator: {dp_info['fallback']}")
        dp_info["usable_comms"] = usable
        if rank == 0:
            print(f"[bench] dp pre-flight: {json.dumps(dp_info['preflight'])} fallback={dp_info['fallback']}", file=sys.stderr, flush=True)
        # candidates of the startup probe: (kind, form); the first one is the fallback of last resort (c10d all-reduce: the most travelled RCCL path)
        if env_form or env_kind or args.no_dp_probe:
            k0 = env_kind or usable[-1]
            cands = [(k0, env_form or ("rs_ag" if k0 == "native" else "allreduce"))]
        else:
            cands = [("torch", "allreduce")] if "torch" in usable else []
            if "native" in usable:
                cands += [("native", "rs_ag"), ("native", "rs_adamw_ag")]
            elif "torch" in usable:
                cands += [("torch", "rs_adamw_ag")]
        configure_dp(*cands[0])
    else:
        cands = []
        opt = FusedAdamW(model.arena, **opt_kw)
        if args.clip > 0:
            opt.clip_norm = args.clip  # norm in one pass over the gradient arena, coefficient applied inside the AdamW launches
        opt.sync_master()
        overlap = None if args.no_opt_overlap else BackwardOverlap(model.arena, opt, None)
        if overlap is not None and os.environ.get("AFK_THIN_BLOCKS"):
            overlap.thin_blocks = int(os.environ["AFK_THIN_BLOCKS"])
    from audio_flamingo_amd import functional as F_
    model.arena.lazy_T_shadows = F_.BWD_FORM == "direct"
    model.arena.refresh_shadows(force=True)
    model.arena.enable_wgrad_stream(not args.no_wgrad_stream)
    model.arena.thin_blocks = int(os.environ.get("AFK_THIN_TRANSPOSE", "0"))
    frontend = LogMelFrontend(dev)
    # the critical path on a stream ABOVE the default queue priority, the wgrad / optimizer streams below it (audio_flamingo_amd/streams.py);
    # AFK_STREAM_PRIORITIES=0 = every stream at the default priority (the round-5 schedule)
    from audio_flamingo_amd import streams as _streams

    compute_stream = None
    if _streams.enabled():
        compute_stream = _streams.make_stream(dev, "compute")
        compute_stream.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(compute_stream)
    # `--batches` distinct synthetic batches live in HBM; every step trains on the next one (copied into the static input tensors the step -
    # and its HIP graph - reads).  Training the SAME batch every step (round 2) drives the loss to ~0 within the timed region: saturated
    # softmax, vanishing gradients - and on a power-limited chip operand statistics move the clock.
    nb = args.batches if args.batches > 0 else min(args.warmup + args.steps + 8, 96)
    pool = [synthetic_batch(args.batch, (rank + k * world) * args.batch, dev, windows) for k in range(nb)]
    waves, ids, labels = (t.clone() for t in pool[0])
    model.label_rows_static = True   # the labelled POSITIONS are the same in every synthetic batch; only the token values change

    data = {"waves": waves, "ids": ids, "labels": labels}
    hbm_probe = {}   # {"events": (start, end)} while the serial profiling step runs: HIP events around its one full-width AdamW launch set
    step_no = [0]

    def load_next():
        """next synthetic batch into the static input tensors (three device-to-device copies, inside the timed region)"""
        if nb > 1:
            w_, i_, l_ = pool[step_no[0] % nb]
            data["waves"].copy_(w_), data["ids"].copy_(i_), data["labels"].copy_(l_)
        step_no[0] += 1

    def step(serial=False):
        feats = frontend(data["waves"], out_dtype=torch.bfloat16)
        model.arena.zero_grad()
        if overlap is not None and not serial:
            # per bucket, inside backward, on a side stream: [all-reduce] -> AdamW -> W^T shadow refresh
            overlap.begin_step()
            out = model(input_ids=data["ids"], input_features=feats, labels=data["labels"])
            out.loss.backward()
            overlap.finish()
            return out.loss
        if engine is not None:
            engine.begin_backward()
        out = model(input_ids=data["ids"], input_features=feats, labels=data["labels"])
        out.loss.backward()
        if engine is not None:
            engine.finish()
        ev = hbm_probe.get("events")
        if ev is not None:
            ev[0].record()
        opt.step(grad_scale=engine.grad_scale if engine is not None else 1.0, gates=engine.bucket_gate if engine is not None else None, refresh_shadows=ev is None)
        if ev is not None:   # the dominant HBM-bound kernel of the step, alone on the chip: 28 B/param
            ev[1].record()
            model.arena.refresh_shadows(force=True)
        return out.loss

    def fence():
        torch.cuda.synchronize()
        if use_dp:
            dist.barrier()
            torch.cuda.synchronize()

    # Default (round 6): the step is enqueued from Python on prioritised streams - the host needs ~41 ms per step on an idle GPU and keeps ahead of the
    # ~390 ms the GPU needs.  --graph (N = 1): the step (static shapes) captured once into a HIP graph - three streams, ~3 000 launches -> one
    # hipGraphLaunch per step.  N > 1 always enqueues eagerly (RCCL inside a capture is validated at world 1 only: --dp-graph).
    want_graph = args.graph or args.dp_graph or (not args.no_graph and not _streams.enabled())
    use_graph = want_graph and (not use_dp or args.dp_graph) and overlap is not None and not ckpt
    load_next()
    first_loss = float(step().detach())  # ~ ln(152064) = 11.9 for random-init weights: the line checks itself (the last loss is lower)
    if use_dp and len(cands) > 1:
        # startup probe: 1 warm-up + 2 timed eager steps per candidate, max over the ranks; the fastest form runs the benchmark.
        # Every rank sees the same (all-reduced) times, so every rank picks the same candidate.  A candidate that raises is dropped - on every rank
        # (the verdict is exchanged) - and the probe goes on; the real steps it costs train the replicas like any other step.
        probe = {}
        for kind, form in cands:
            name = f"{kind}:{form}"
            ok_here = 1
            try:
                if (kind, form) != cands[0] or engine is None:
                    configure_dp(kind, form)
                load_next()
                step()
                fence()
                t_p = time.perf_counter()
                for _ in range(2):
                    load_next()
                    step()
                fence()
                ms_p = 1e3 * (time.perf_counter() - t_p) / 2
            except Exception as e:
                ok_here, ms_p = 0, 1e9
                dp_info["fallback"].append(f"probe {name}: {type(e).__name__}: {str(e)[:160]}")
            tt = torch.tensor([ms_p, float(1 - ok_here)], device=coll_dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            if float(tt[1]) == 0.0:
                probe[name] = round(float(tt[0]), 2)
        assert probe, f"every exchange form failed its probe: {dp_info['fallback']}"
        best = min(probe, key=probe.get)
        dp_info["probe_ms"], dp_info["chosen"] = probe, best
        if rank == 0:
            print(f"[bench] dp form probe (ms/step, max over ranks): {probe} -> {best}", file=sys.stderr, flush=True)
        bk, bf = best.split(":")
        configure_dp(bk, bf)   # fresh optimizer state for the chosen form (the probe's steps trained the replicas like any other step)
    elif use_dp:
        dp_info["chosen"] = f"{cands[0][0]}:{cands[0][1]}"
    if overlap is not None and not use_graph and use_dp:
        overlap.measure_tail = True
    run = step
    if use_graph:
        from audio_flamingo_amd.graphs import GraphedTrainStep

        run = GraphedTrainStep(model, opt, overlap, step, warmup=1, stream=compute_stream)
    loss = None
    for i in range(args.warmup):
        load_next()
        loss = run()
    fence()
    ops.prof_reset()
    ops.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        load_next()
        loss = run()
    host_enqueue = time.perf_counter() - t0  # host time to enqueue the steps (no sync inside; includes queue back-pressure once the GPU lags)
    fence()
    dt = time.perf_counter() - t0
    if overlap is not None and overlap.measure_tail:
        tail = overlap.exposed_tail_ms(last=args.steps)   # compute stream idle behind the last backward kernel: [exchange ->] AdamW -> shadows of the last buckets
        if use_dp:
            tt = torch.tensor([tail or 0.0], device=coll_dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tail = float(tt.item())
        dp_info["exposed_comm_ms"] = None if tail is None else round(tail, 2)
        overlap.measure_tail = False
    # the same WITHOUT back-pressure: one more (untimed) step enqueued onto an idle GPU - what the host really needs per step
    t1 = time.perf_counter()
    load_next()
    loss = run()
    host_enqueue_idle = time.perf_counter() - t1
    fence()
    ops.prof_enable(False)
    ov_ms, ov_flops, ov_launches = ops.prof_collect()
    # Per-launch GEMM durations: with the wgrad branch on a second stream and AdamW on a third, kernels share the chip and the
    # HIP-event brackets of the timed region over-state each launch.  One extra UNTIMED step on the serial schedule (one stream,
    # optimizer after backward) gives the same launches back to back; that is what `roofline.achieved` is computed from (both are
    # reported; profiles/r01f_bench_kernel_stats_serial.md is the rocprofv3 view of the same schedule).
    had_side = model.arena.wgrad_stream is not None
    if had_side:
        model.arena.join_streams()
        model.arena.enable_wgrad_stream(False)
        fence()
        ops.prof_reset()
        ops.prof_enable(True)
        hbm_probe["events"] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        step(serial=True)
        fence()
        ops.prof_enable(False)
        adamw_ms = hbm_probe["events"][0].elapsed_time(hbm_probe["events"][1])
        hbm_probe.clear()
        gemm_ms, gemm_flops, gemm_launches = ops.prof_collect()
        prof_steps = 1
        if os.environ.get("AFK_PROF_DUMP"):
            from audio_flamingo_amd import _lib
            _lib.call("afk_prof_dump", os.environ["AFK_PROF_DUMP"].encode())
    else:
        adamw_ms = None
        gemm_ms, gemm_flops, gemm_launches, prof_steps = ov_ms, ov_flops, ov_launches, args.steps
    final_loss = float(loss.detach()) if loss is not None else float("nan")
    if first_loss is None:
        first_loss = final_loss
    rank_losses, replicas_identical = [final_loss], None
    if use_dp:
        t = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # every replica must hold bit-identical parameters after the timed steps (same reduced gradients, same AdamW): checksum per rank
        model.arena.join_streams()
        p32 = model.arena.params.float()
        chk = torch.stack([p32.sum().double(), p32.abs().sum().double(), torch.tensor(final_loss, device=dev, dtype=torch.float64)]).to(coll_dev)
        allc = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        replicas_identical = all(bool(torch.equal(c[:2], allc[0][:2])) for c in allc)
        rank_losses = [float(c[2]) for c in allc]
        assert replicas_identical, f"data-parallel replicas diverged: {[c[:2].tolist() for c in allc]}"
        del p32
    peak_mem = torch.cuda.max_memory_allocated(dev) / 2 ** 30

    # parameter checksum of this replica after the timed steps (every run prints it: two schedules of the same code - eager / HIP graph, with /
    # without the RCCL path - must agree on it bit for bit, tests/test_dp_gpu.py)
    model.arena.join_streams()
    torch.cuda.synchronize()
    _p32 = model.arena.params.float()
    param_checksum = [float(_p32.sum().double()), float(_p32.abs().sum().double())]
    del _p32

    # BASELINE configs[4] (long audio) beside the headline: a few steps of the 5-minute workload on the same replica, so that the
    # driver's plain `bench.py --gpus N` run records it too (one sample = 10 windows, S = 7 774, per-layer activation checkpointing);
    # then the same at AF3's stated maximum clip length (10 minutes = 20 windows, S = 15 274; /root/reference README.md:109)
    def long_leg(name, label):
        """both recompute plans of the long-audio workloads: "reference" = every layer recomputed, what
        gradient_checkpointing_enable() means in the reference (modeling_layers.py:79-114) and by default here; "budgeted" = the opt-in plan that
        recomputes only what does not fit 0.85 x HBM.  The leg's headline fields are the reference-semantics run; `budgeted_plan` sits beside it."""
        lw = WORKLOADS[name]
        ls = 9 + N_AUDIO_TOK * lw["windows"] + 9 + N_ANSWER
        data["waves"], data["ids"], data["labels"] = synthetic_batch(lw["batch"], rank * lw["batch"], dev, lw["windows"])
        runs = {}
        for pol in ("full", "budget"):
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats(dev)
            model.gradient_checkpointing_enable(dict(policy=pol))
            for _ in range(2):
                step()
            fence()
            t0 = time.perf_counter()
            for _ in range(3):
                step()
            fence()
            ldt = time.perf_counter() - t0
            if use_dp:
                t = torch.tensor([ldt], device=coll_dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ldt = float(t.item())
            model.gradient_checkpointing_disable()
            plan = {k: v for k, v in (model.ckpt_plan or {}).items() if not k.startswith("_")}
            sps = world * lw["batch"] * 3 / ldt
            runs[pol] = {"ms_per_step": 1000.0 * ldt / 3, "steps": 3, "warmup": 2,
                         "value": sps * CLIP_SECONDS * lw["windows"], "unit": "audio-s/s", "decoder_tokens_per_s": sps * ls,
                         "model_tflops_per_gpu": train_flops_per_sample(ls, lw["windows"]) * sps / world / 1e12,
                         "hardware_tflops_per_gpu": train_flops_per_sample(ls, lw["windows"], (plan.get("enc", 32), plan.get("dec", 28))) * sps / world / 1e12,
                         "layers_recomputed": {"encoder": f"{plan.get('enc')}/32", "decoder": f"{plan.get('dec')}/28"},
                         "checkpoint_plan": plan, "peak_mem_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1)}
        res_ = {"workload": f"AF3-7B bf16 train step, {label} = {lw['windows']} windows/sample, S={ls}, micro-batch {lw['batch']}/GPU, activation checkpointing ON "
                            f"as the reference runs it (EVERY layer of both towers recomputed, modeling_layers.py:79-114); `budgeted_plan` = the opt-in "
                            f"memory-budgeted plan (policy 'budget') on the same batch"}
        res_.update(runs["full"])
        res_["budgeted_plan"] = runs["budget"]
        data["waves"], data["ids"], data["labels"] = waves, ids, labels
        return res_

    # CU-contention sweep (before the legs below change the model's mode): the graphed / overlapped step beside n parked workgroups
    cu_contention = None
    if args.cu_contention and not use_dp:
        from audio_flamingo_amd import _lib
        cu_contention = {}
        hog_stream = torch.cuda.Stream(device=dev)
        flag = torch.zeros(1, dtype=torch.int32).pin_memory()   # host-coherent stop flag: the CPU ends the parked kernel with a plain store
        for n_hog in [int(x) for x in args.cu_contention.split(",") if x.strip() != ""]:
            flag[0] = 0
            fence()
            rep = torch.zeros((max(n_hog, 1), 3), device=dev, dtype=torch.int64)
            if n_hog > 0:   # 96 KiB of LDS per parked workgroup: ONE per CU, and a 128 KiB GEMM workgroup cannot share that CU; at most 30 s (100 MHz ticks)
                # bounded life: 0.9 s per timed step + settle (100 MHz ticks) - the stop flag is best effort (the parked waves poll it with system-scope
                # acquire loads; on this stack they did not observe the host's store and left at the bound, which is why the bound is tight)
                ticks = int((0.9 * 6 + 0.6) * 1e8)
                _lib.call("afk_cu_hog", n_hog, 96 * 1024, flag.data_ptr(), ticks, rep.data_ptr(), hog_stream.cuda_stream)
                time.sleep(0.05)   # let the parked workgroups land before the step's kernels arrive
            load_next()
            run()                      # one step for the parked workgroups to settle on their CUs
            torch.cuda.current_stream().synchronize()
            t_h = time.perf_counter()
            for _ in range(5):
                load_next()
                run()
            torch.cuda.current_stream().synchronize()
            if overlap is not None:
                model.arena.join_streams()
                torch.cuda.current_stream().synchronize()
            cu_contention[str(n_hog)] = round(1000.0 * (time.perf_counter() - t_h) / 5, 2)
            t_steps_ms = 1000.0 * (time.perf_counter() - t_h)
            flag[0] = 1                # host store into pinned memory: the parked workgroups see it through their system-scope acquire loads
            fence()
            if n_hog > 0:              # residency proof: how long the parked workgroups lived (they must span the timed steps) and on how many distinct CUs
                r = rep.cpu()
                alive_ms = (r[:, 1].float() / 1e5).tolist()
                cus = len({(int(v) >> 32, (int(v) >> 8) & 0xF, (int(v) >> 13) & 0x7) for v in r[:, 2].tolist()})   # (XCC, CU_ID bits 11:8, SE_ID bits 15:13)
                cu_contention[f"{n_hog}_parked_ms_min_max"] = [round(min(alive_ms), 1), round(max(alive_ms), 1)]   # must exceed the timed steps:
                cu_contention[f"{n_hog}_timed_steps_ms"] = round(t_steps_ms, 1)
                cu_contention[f"{n_hog}_distinct_cus"] = cus
        print(f"[bench] cu_contention (ms/step beside n parked CUs): {cu_contention}", file=sys.stderr, flush=True)

    long_audio = long_10min = None
    if args.workload == "clip30" and full_model and not args.no_long_audio:
        long_audio = long_leg("long5min", "5-min clips (BASELINE configs[4])")
        long_10min = long_leg("long10min", "10-min clips (AF3's stated maximum, /root/reference README.md:109)")

    # KV-cache decode on the same replica: generate() from a 768-token prompt (9 + 750 <sound> + 9), B = 1 and B = 8;
    # per-token time = (33 new tokens - 1 new token) / 32, i.e. the graph-replayed decode step incl. its one-time capture, against the
    # weight-streaming roofline (every decoder weight + lm_head read once per token: 15.2 GB bf16 at 8 TB/s)
    decode = None
    if args.workload == "clip30" and full_model and not args.no_extra_legs and world == 1:
        try:
            decode = decode_leg(model, frontend, waves, ids, dev)
        except Exception as e:
            decode = {"error": repr(e)[:300]}

    if rank == 0:
        ms_per_step = 1000.0 * dt / args.steps
        samples_per_s = world * args.batch * args.steps / dt
        achieved_tf = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        # roofline.traffic: PMC passes cannot run inside bench.py, so the committed rocprofv3 --pmc result (tools/measure_gemm_traffic.py) is
        # quoted - but ONLY when it was measured on THIS build of the kernels (afk_build_id stamp); a figure from other sources is refused
        import glob

        from audio_flamingo_amd import _lib as _afk_lib

        build_id = _afk_lib.load().afk_build_id().decode()
        traffic, traffic_detail = None, {"this_build": build_id, "note": "no PMC measurement of this build under profiles/ (python tools/measure_gemm_traffic.py)"}
        for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_traffic*.json")), reverse=True):
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("afk_build_id") == build_id:
                traffic = tj["hbm_bytes_per_launch"]  # HBM bytes of ONE launch of the kernel on the shape below (largest GEMM of the step)
                traffic_detail = {"hbm_bytes_per_launch": tj["hbm_bytes_per_launch"], "algorithmic_bytes_per_launch": tj["algorithmic_bytes_per_launch"],
                                  "shape": tj["shape"], "afk_build_id": build_id, "l2_hit_rate": tj.get("l2_hit_rate"),
                                  "source": os.path.relpath(tpath, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; stamped with the build it ran on)"}
                break
            traffic_detail.setdefault("stale", []).append({"file": os.path.relpath(tpath, ROOT), "measured_on_build": tj.get("afk_build_id", "unstamped (round 2)"),
                                                            "hbm_bytes_per_launch": tj.get("hbm_bytes_per_launch")})
        model_tf = train_flops_per_sample(s_tok, windows) * samples_per_s / world / 1e12 if full_model else None
        plan = {k: v for k, v in (model.ckpt_plan or {}).items() if not k.startswith("_")} if ckpt else None
        rc = (plan.get("enc", 32), plan.get("dec", 28)) if plan else bool(ckpt)     # layers the memory-budgeted plan really recomputed
        hw_tf = train_flops_per_sample(s_tok, windows, rc) * samples_per_s / world / 1e12 if full_model else None
        # lm_head + loss (forward, dgrad, wgrad) run only on the rows that carry a label: identical loss and gradients, fewer executed FLOPs.
        # model_tflops stays ALGORITHMIC (the reference's 3 x forward over every row); executed = what the kernels really did.
        skipped = 3.0 * 2 * (s_tok - N_ANSWER) * 3584 * 152064 if model.loss_on_valid_rows_only else 0.0
        if model.loss_on_valid_rows_only and getattr(model, "last_layer_rows_only", False):
            # round 6: behind its attention the LAST decoder layer (o_proj, MLP) runs on the labelled rows only as well (same loss, same gradients)
            skipped += 3.0 * 2 * (s_tok - N_ANSWER) * (3584 * 3584 + 3 * 3584 * 18944)
        exec_tf = (train_flops_per_sample(s_tok, windows, rc) - skipped) * samples_per_s / world / 1e12 if full_model else None
        res = {
            "metric": "audio-sec/s + decoder tokens/s, AF3-7B bf16 train",
            "value": samples_per_s * CLIP_SECONDS * windows, "unit": "audio-s/s",
            "decoder_tokens_per_s": samples_per_s * s_tok, "answer_tokens_per_s": samples_per_s * N_ANSWER,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ((("AF3 (AF-Whisper 32L + Qwen2.5-7B 28L, MLP projector) bf16 train step fwd+bwd+AdamW, 30 s clips, S=1024" if windows == 1 else
                                      f"AF3-7B bf16 train step fwd+bwd+AdamW, LONG AUDIO: 5-min clips = {windows} windows/sample, S={s_tok}, "
                                      f"activation checkpointing {'ON (plan: ' + str(plan) + ')' if ckpt else 'OFF'} (BASELINE configs[4])")) if full_model
                                    else f"DEPTH-REDUCED AF3 ({args.enc_layers} enc + {args.dec_layers} dec layers) - not the BASELINE config"),
                       "micro_batch_per_gpu": args.batch, "global_batch": args.batch * world, "seq_len": s_tok, "audio_tokens": n_audio_tok,
                       "windows_per_sample": windows, "activation_checkpointing": ckpt, "checkpoint_plan": plan, "max_grad_norm": args.clip if args.clip > 0 else None,
                       "parallelism": f"dp{world}", "params": model.trainable_numel()},
            "step_enqueue": "hip_graph_replay" if use_graph else "eager_python",
            "stream_priorities": None if compute_stream is None else {"range_least_greatest": list(_streams.priority_range()), "compute": compute_stream.afk_priority,
                                                                      "wgrad": getattr(model.arena.wgrad_stream, "afk_priority", None),
                                                                      "side": getattr(getattr(overlap, "side", None), "afk_priority", None),
                                                                      "side_cus": getattr(getattr(overlap, "side", None), "afk_cus", None)},
            **({"INVALID_probe": "AFK_PROBE_SKIP_ADAMW=1: the optimizer launches were skipped (timing probe)"} if os.environ.get("AFK_PROBE_SKIP_ADAMW", "0") == "1" else {}),
            "loss": final_loss, "loss_first_step": first_loss, "rank_losses": rank_losses, "rccl_ranks": world if use_dp else 0,
            "replicas_identical_after_steps": replicas_identical, "param_checksum": param_checksum, "synthetic_batches_rotated": nb, "label_rows_static": bool(model.label_rows_static),
            "dp": None if engine is None else {**dp_info, "backend": args.backend, "comm": engine.comm_kind if engine.native is not None else "torch",
                                               "form": engine.form if (engine.native is not None or engine.sharded) else "allreduce",
                                               "optimizer": ("sharded over the ranks (reduce-scatter -> AdamW on 1 / world of every bucket -> all-gather of the bf16 parameters)"
                                                             if engine.sharded else "replicated"),
                                               "optimizer_state_gib": round(12 * getattr(opt, "state_numel", model.arena.total) / 2 ** 30, 2),
                                               "collectives_forced_at_world_1": bool(engine.force_collectives and world == 1),
                                               "buckets": len(model.arena.bucket_names), "bucket_bytes_max": 2 * max(e - s_ for s_, e in model.arena._bucket_ranges),
                                               "bucket_bytes_total": 2 * model.arena.total, "overlapped_with_backward": overlap is not None or engine.overlap,
                                               "collectives_per_step": len(model.arena.bucket_names) + 1,
                                               "rccl_channel_env": {k: os.environ.get(k) for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS", "RCCL_MSCCL_ENABLE") if os.environ.get(k) is not None},
                                               },
            "long_audio_configs4": long_audio, "long_audio_10min": long_10min, "decode": decode, "cu_contention_ms_per_step": cu_contention,
            "peak_mem_gib": round(peak_mem, 1), "host_enqueue_ms_per_step": round(1000.0 * host_enqueue / args.steps, 1),
            "host_enqueue_ms_idle_gpu": round(1000.0 * host_enqueue_idle, 1),
            "waited_for_free_hbm_s": waited,
            "model_tflops_per_gpu": model_tf, "model_frac_of_mfma_peak": (model_tf / 2500.0) if model_tf else None,
            "hardware_tflops_per_gpu": hw_tf, "executed_tflops_per_gpu": exec_tf,
            "roofline_hbm": None if not adamw_ms else {
                "bound": "hbm", "kernel": "adamw_kernel (fused AdamW over the flat arena: bf16 grad + fp32 master / m / v read, master / m / v + bf16 param written = 28 B/param)",
                "achieved": 28.0 * model.trainable_numel() / (adamw_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                "frac": 28.0 * model.trainable_numel() / (adamw_ms * 1e-3) / 1e9 / 8000.0, "ms": adamw_ms, "algorithmic_bytes": 28.0 * model.trainable_numel(),
                "note": "HIP events around the full-width AdamW launches of the extra serial step (alone on the chip); 6.29 TB/s is the achievable streaming rate (MI355X_MICROARCH.md)"},
            "lm_head_rows": {"executed": N_ANSWER, "of": s_tok, "note": "lm_head/CE and their backward GEMMs run on the labelled rows only (same loss, same gradients)",
                             "last_decoder_layer_behind_its_attention": bool(getattr(model, "last_layer_rows_only", False))},
            "roofline": {"bound": "mfma", "kernel": "gemm_nt_bf16_k256 (+ k128 for <192-tile shapes): every dense contraction (fwd, dgrad, wgrad, conv stem, lm_head)",
                         "achieved": achieved_tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": achieved_tf / 2500.0, "traffic": traffic, "traffic_detail": traffic_detail,
                         "launches": gemm_launches, "avg_launch_ms": gemm_ms / max(gemm_launches, 1),
                         "gemm_ms_per_step": gemm_ms / prof_steps,
                         # the timed region's own GEMM figure.  Eager enqueue: the same HIP-event brackets inside the timed steps (inflated when streams share
                         # the chip).  HIP-graph replay: events recorded during capture carry no timestamps, so per-launch times do not exist there - what
                         # exists is a FLOOR: the GEMM flops one step executes (counted on the serial step) over the measured step time, i.e. the rate the
                         # GEMMs sustain while sharing the step with every other kernel
                         "timed_region_overlapped": {"achieved": (ov_flops / (ov_ms * 1e-3) / 1e12) if ov_ms > 0 else None, "launches": ov_launches,
                                                     "gemm_tflops_over_whole_step": (gemm_flops / prof_steps) / (ms_per_step * 1e-3) / 1e12 if gemm_flops > 0 else None,
                                                     "frac_over_whole_step": (gemm_flops / prof_steps) / (ms_per_step * 1e-3) / 1e12 / 2500.0 if gemm_flops > 0 else None,
                                                     "note": "achieved = per-launch HIP events inside the timed steps (null under HIP-graph replay: captured events carry no "
                                                             "timestamps); gemm_tflops_over_whole_step = executed GEMM flops of one step / measured ms_per_step - a floor for the "
                                                             "GEMMs' rate inside the overlapped timed region"},
                         "note": ("HIP events around every GEMM launch on its launch stream; measured on one extra untimed step with the wgrad stream "
                                  "disabled (launches back to back)" if had_side else "HIP events around every GEMM launch, timed region")
                                 + "; the gate|up launches carry the fused SwiGLU forward: its elementwise work counts as GEMM time, not as flops"},
        }
        if world == 1 and full_model and not args.no_extra_legs:
            # the measured ceiling the GEMM is held against: pure-MFMA rate of this chip under its power cap, live, beside `frac`
            try:
                ceil_tf = mfma_power_ceiling(dev)
                res["roofline"]["power_ceiling_tflops"] = ceil_tf
                res["roofline"]["frac_of_power_ceiling"] = achieved_tf / ceil_tf
            except Exception as e:
                res["roofline"]["power_ceiling_error"] = repr(e)[:200]
        want_eager = not args.no_eager_baseline and world == 1 and args.workload == "clip30"
        want_icl = not args.no_extra_legs and world == 1 and args.workload == "clip30" and full_model
        want_parity = not args.no_parity and world == 1 and args.workload == "clip30" and full_model and args.batch == 8
        if want_eager or want_icl or want_parity:
            # both need the HBM of our replica (the ICL model: 4.3 B parameters + fp32 AdamW state; the reference model: 125 GiB): free it first
            feats_b = frontend(waves, out_dtype=torch.bfloat16)
            model.arena.on_bucket_ready = None
            loss = None
            data.clear()
            pool.clear()
            del model, opt, overlap, engine, step, run, long_leg
            import gc

            gc.collect()
            torch.cuda.empty_cache()
        if want_parity:
            # UNTIMED: the configuration timed above against the live reference (fp32 = truth, bf16 = noise floor) with ONE shared state_dict
            try:
                from tools import parity_fulldepth as _pf

                rec = _pf.run(dev)
                _pf.write_record(rec)
                res["parity_fulldepth"] = _pf.summary(rec)
                del rec
                gc.collect()
                torch.cuda.empty_cache()
            except Exception as e:
                res["parity_fulldepth"] = {"green": None, "error": repr(e)[:400]}
        if want_icl:
            # BASELINE configs[3] (AF1/AF2-style ICL step) in the driver-visible line: builder-declared shapes, parity UNPINNED (no AF1/AF2 code
            # exists in the mount: SURVEY.md §0) - see run_icl4
            try:
                torch.cuda.reset_peak_memory_stats(dev)
                ia = argparse.Namespace(batch=WORKLOADS["icl4"]["batch"], dec_layers=28, warmup=2, steps=5)
                r4 = run_icl4(ia, dev)
                res["icl4_configs3"] = {k: r4[k] for k in ("metric", "value", "unit", "decoder_tokens_per_s", "ms_per_step", "steps", "warmup", "loss", "loss_first_step",
                                                           "peak_mem_gib", "config")}
                res["icl4_configs3"]["roofline_gemm_frac"] = r4["roofline"]["frac"]
                res["icl4_configs3"]["parity"] = "UNPINNED (stand-in oracle: Idefics blocks; oracle/flamingo_oracle.py)"
                gc.collect()
                torch.cuda.empty_cache()
            except Exception as e:
                res["icl4_configs3"] = {"value": None, "error": repr(e)[:300]}
        if want_eager:
            # the reference's own model on this GPU
            try:
                torch.cuda.reset_peak_memory_stats(dev)
                still = torch.cuda.memory_allocated(dev) / 2 ** 30
                eb = eager_rocm_baseline(dev, feats_b, ids, labels, clip=args.clip)
                eb["hbm_still_allocated_before_gib"] = round(still, 1)
                res["eager_rocm_baseline"] = eb
                if eb.get("value"):
                    res["speedup_vs_eager_rocm"] = res["value"] / eb["value"]
            except Exception as e:
                res["eager_rocm_baseline"] = {"value": None, "unit": "audio-s/s", "error": repr(e)[:300]}
        if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N = 1 only
            try:
                res["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the baseline must never take the measured line down
                res["cpu_baseline"] = {"value": None, "unit": "audio-s/s", "cores": os.cpu_count(), "kind": "reference", "sample": f"failed: {e!r}"}
        # hardware = executed (what the kernels really did, recompute included, unlabelled lm_head rows excluded); the algorithmic figure stays `model_*`
        res["hardware_tflops_per_gpu"] = res["executed_tflops_per_gpu"]
        res["executed_frac_of_mfma_peak"] = (res["executed_tflops_per_gpu"] / 2500.0) if res["executed_tflops_per_gpu"] else None
        line = compact_line(res, write_detail(res, args.detail_name))
    else:
        line = None
    def drain():
        # RCCL writes its version banner through C stdio (fully buffered on a pipe -> it would surface at process exit, AFTER the
        # result): every rank drains both layers
        sys.stdout.flush()
        try:
            import ctypes

            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass

    drain()
    if use_dp:
        dist.barrier()
        torch.cuda.synchronize()
        if use_graph:   # a captured graph keeps the communicator's kernels and buffers referenced: let it go BEFORE the communicator is destroyed (the 1-rank
            run.graph.reset()   # RCCL graph run of tests/test_dp_gpu.py died with SIGABRT from a c10 worker thread in 3 of 11 full-suite runs; teardown order is the suspect)
            run = None
            torch.cuda.synchronize()
        dist.destroy_process_group()
        drain()
    if line is not None:
        if world > 1:
            time.sleep(2.0)  # the other ranks have nothing left to print, let them leave: the JSON line stays the last line of output
        print(line, flush=True)


if __name__ == "__main__":
    main()
