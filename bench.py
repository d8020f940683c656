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
WORKLOADS = {"clip30": dict(windows=1, batch=8, checkpoint=False), "long5min": dict(windows=10, batch=1, checkpoint=True)}


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
    """algorithmic FLOPs (2*MACs, causal attention at 1/2), forward x 3; recompute=True adds the checkpointed layers' second
    forward (x 4 on the layer stacks) - reported separately as hardware FLOPs, never as model FLOPs"""
    enc_l = 32 * (2 * 1500 * (4 * 1280 ** 2 + 2 * 1280 * 5120) + 4 * 1500 ** 2 * 1280)
    enc = windows * (2 * 3000 * 128 * 3 * 1280 + 2 * 1500 * 1280 * 3 * 1280)
    proj = windows * 2 * 750 * (1280 * 3584 + 3584 * 3584)
    dec = 28 * (2 * S * (2 * 3584 ** 2 + 2 * 3584 * 512 + 3 * 3584 * 18944) + 2 * S ** 2 * 3584)
    lm = 2 * S * 3584 * 152064
    layers = windows * enc_l + dec
    return 3.0 * (enc + proj + lm) + (4.0 if recompute else 3.0) * layers


def cpu_baseline(seconds_budget=30.0):
    """The oracle (CPU restatement of the reference algorithm, fp32) timed on this host's cores on a bounded sample:
    one sample (1 window, S=1024) through ONE encoder layer (+stem), ONE decoder layer and lm_head+CE, forward+backward,
    composed to the full 32+28-layer step.  A reported baseline, not a target."""
    from oracle import af3_oracle as O

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, generator=g) * 0.02)
    E, Fd, H, I, V = 1280, 5120, 3584, 18944, 152064
    at, lm = "model.audio_tower.", "model.language_model."
    sd = {at + "conv1.weight": rnd(E, 128, 3), at + "conv1.bias": torch.zeros(E), at + "conv2.weight": rnd(E, E, 3), at + "conv2.bias": torch.zeros(E),
          at + "embed_positions.weight": rnd(1500, E), at + "layer_norm.weight": torch.ones(E), at + "layer_norm.bias": torch.zeros(E)}
    p = at + "layers.0."
    for n, shp in (("self_attn.q_proj", (E, E)), ("self_attn.k_proj", (E, E)), ("self_attn.v_proj", (E, E)), ("self_attn.out_proj", (E, E)), ("fc1", (Fd, E)), ("fc2", (E, Fd))):
        sd[p + n + ".weight"] = rnd(*shp)
        if n != "self_attn.k_proj":
            sd[p + n + ".bias"] = torch.zeros(shp[0])
    for n in ("self_attn_layer_norm", "final_layer_norm"):
        sd[p + n + ".weight"], sd[p + n + ".bias"] = torch.ones(E), torch.zeros(E)
    q = lm + "layers.0."
    for n, shp in (("self_attn.q_proj", (H, H)), ("self_attn.k_proj", (512, H)), ("self_attn.v_proj", (512, H))):
        sd[q + n + ".weight"], sd[q + n + ".bias"] = rnd(*shp), torch.zeros(shp[0])
    sd[q + "self_attn.o_proj.weight"] = rnd(H, H)
    sd[q + "mlp.gate_proj.weight"], sd[q + "mlp.up_proj.weight"], sd[q + "mlp.down_proj.weight"] = rnd(I, H), rnd(I, H), rnd(H, I)
    for n in ("input_layernorm", "post_attention_layernorm"):
        sd[q + n + ".weight"] = torch.ones(H)
    sd[lm + "norm.weight"] = torch.ones(H)
    head = rnd(V, H).requires_grad_(True)
    for v in sd.values():
        v.requires_grad_(True)

    def timed(fn):
        t0 = time.perf_counter()
        out = fn()
        out.backward()
        return time.perf_counter() - t0

    feats = torch.randn(1, 128, 3000, generator=g)
    x_dec = torch.randn(1, S_TOK, H, generator=g).requires_grad_(True)
    labels = torch.randint(0, V, (S_TOK,), generator=g)
    labels[: S_TOK - N_ANSWER] = -100
    t_enc_all = timed(lambda: O.encoder(sd, feats, None, 20)[0].float().pow(2).mean())          # stem + 1 layer + pool/LN
    t_dec = timed(lambda: O.decoder(sd, x_dec, 28, 4, 1e-6, 1e6).float().pow(2).mean())          # 1 layer + final norm
    t_head = timed(lambda: torch.nn.functional.cross_entropy(torch.nn.functional.linear(x_dec[0], head).float(), labels, ignore_index=-100))
    # isolate the per-layer encoder cost with a stem-only run
    sd0 = {k: v for k, v in sd.items() if ".layers.0." not in k or not k.startswith(at)}
    t_stem = timed(lambda: O.encoder(sd0, feats, None, 20)[0].float().pow(2).mean())
    t_enc_layer = max(t_enc_all - t_stem, 1e-6)
    full = t_stem + 32 * t_enc_layer + 28 * t_dec + t_head
    return {
        "value": CLIP_SECONDS / full, "unit": "audio-s/s", "cores": cores, "kind": "port",
        "decoder_tokens_per_s": S_TOK / full,
        "sample": (f"oracle/af3_oracle.py fp32, B=1 (one 30 s window, S=1024) fwd+bwd of stem ({t_stem:.2f}s), 1 encoder layer ({t_enc_layer:.2f}s), "
                   f"1 decoder layer ({t_dec:.2f}s), lm_head+CE ({t_head:.2f}s); composed to 32 enc + 28 dec layers = {full:.1f}s/sample (extrapolated, no optimizer)"),
    }


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
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="clip30", help="clip30 = BASELINE configs[1]/[2] (the headline); long5min = configs[4]")
    ap.add_argument("--batch", type=int, default=0, help="micro-batch (samples) per GPU; default 8 for clip30 (BASELINE config), 1 for long5min")
    ap.add_argument("--no-checkpoint", action="store_true", help="long5min only: keep all activations instead of per-layer recompute")
    ap.add_argument("--enc-layers", type=int, default=32)
    ap.add_argument("--dec-layers", type=int, default=28)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--no-opt-overlap", action="store_true", help="run AdamW after backward instead of bucket-by-bucket inside it")
    ap.add_argument("--no-wgrad-stream", action="store_true", help="keep the weight-gradient branch on the compute stream")
    ap.add_argument("--force-dp", action="store_true", help="run the data-parallel engine even with one rank (exercises the RCCL path)")
    args = ap.parse_args()
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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    waited = settle_hbm(dev)
    use_dp = world > 1 or args.force_dp
    if use_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    full_model = args.enc_layers == 32 and args.dec_layers == 28
    model = AudioFlamingo3ForConditionalGeneration(af3_7b_config(args.enc_layers, args.dec_layers), device=dev, init_seed=0)
    model.check_placeholders = False  # the count assertion is a host sync; shapes are static in this benchmark
    if ckpt:
        model.gradient_checkpointing_enable()
    opt = FusedAdamW(model.arena, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    engine = None
    if use_dp:
        engine = DataParallelEngine(model.arena, overlap=not args.no_overlap)
        engine.broadcast_parameters(0)
        opt.master.copy_(model.arena.params)
    from audio_flamingo_amd import functional as F_
    model.arena.lazy_T_shadows = F_.BWD_FORM == "direct"
    model.arena.refresh_shadows(force=True)
    model.arena.enable_wgrad_stream(not args.no_wgrad_stream)
    model.arena.thin_blocks = int(os.environ.get("AFK_THIN_TRANSPOSE", "0"))
    frontend = LogMelFrontend(dev)
    waves, ids, labels = synthetic_batch(args.batch, rank * args.batch, dev, windows)

    overlap = None if args.no_opt_overlap else BackwardOverlap(model.arena, opt, engine)
    if overlap is not None and os.environ.get("AFK_THIN_BLOCKS"):
        overlap.thin_blocks = int(os.environ["AFK_THIN_BLOCKS"])

    def step(serial=False):
        feats = frontend(waves, out_dtype=torch.bfloat16)
        model.arena.zero_grad()
        if overlap is not None and not serial:
            # per bucket, inside backward, on a side stream: [all-reduce] -> AdamW -> W^T shadow refresh
            overlap.begin_step()
            out = model(input_ids=ids, input_features=feats, labels=labels)
            out.loss.backward()
            overlap.finish()
            return out.loss
        if engine is not None:
            engine.begin_backward()
        out = model(input_ids=ids, input_features=feats, labels=labels)
        out.loss.backward()
        if engine is not None:
            engine.finish()
        opt.step(grad_scale=engine.grad_scale if engine is not None else 1.0)
        return out.loss

    def fence():
        torch.cuda.synchronize()
        if use_dp:
            dist.barrier()
            torch.cuda.synchronize()

    loss = None
    for _ in range(args.warmup):
        loss = step()
    fence()
    ops.prof_reset()
    ops.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host_enqueue = time.perf_counter() - t0  # host time to enqueue the steps (no sync inside): << dt means the GPU is never starved
    fence()
    dt = time.perf_counter() - t0
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
        step(serial=True)
        fence()
        ops.prof_enable(False)
        gemm_ms, gemm_flops, gemm_launches = ops.prof_collect()
        prof_steps = 1
        if os.environ.get("AFK_PROF_DUMP"):
            from audio_flamingo_amd import _lib
            _lib.call("afk_prof_dump", os.environ["AFK_PROF_DUMP"].encode())
    else:
        gemm_ms, gemm_flops, gemm_launches, prof_steps = ov_ms, ov_flops, ov_launches, args.steps
    if use_dp:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.detach()) if loss is not None else float("nan")
    peak_mem = torch.cuda.max_memory_allocated(dev) / 2 ** 30

    if rank == 0:
        ms_per_step = 1000.0 * dt / args.steps
        samples_per_s = world * args.batch * args.steps / dt
        achieved_tf = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        traffic, traffic_detail = None, None
        tpath = os.path.join(ROOT, "profiles", "r01_gemm_traffic.json")
        if os.path.exists(tpath):  # PMC passes cannot run inside bench.py; the committed rocprofv3 --pmc result is quoted
            with open(tpath) as f:
                tj = json.load(f)
            traffic = tj["hbm_bytes_per_launch"]  # HBM bytes of ONE launch of the kernel on the shape below (largest GEMM of the step)
            traffic_detail = {"hbm_bytes_per_launch": tj["hbm_bytes_per_launch"], "algorithmic_bytes_per_launch": tj["algorithmic_bytes_per_launch"],
                              "shape": tj["shape"], "source": "profiles/r01_gemm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"}
        model_tf = train_flops_per_sample(s_tok, windows) * samples_per_s / world / 1e12 if full_model else None
        hw_tf = train_flops_per_sample(s_tok, windows, ckpt) * samples_per_s / world / 1e12 if full_model else None
        res = {
            "metric": "audio-sec/s + decoder tokens/s, AF3-7B bf16 train",
            "value": samples_per_s * CLIP_SECONDS * windows, "unit": "audio-s/s",
            "decoder_tokens_per_s": samples_per_s * s_tok, "answer_tokens_per_s": samples_per_s * N_ANSWER,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ((("AF3 (AF-Whisper 32L + Qwen2.5-7B 28L, MLP projector) bf16 train step fwd+bwd+AdamW, 30 s clips, S=1024" if windows == 1 else
                                      f"AF3-7B bf16 train step fwd+bwd+AdamW, LONG AUDIO: 5-min clips = {windows} windows/sample, S={s_tok}, "
                                      f"per-layer activation checkpointing {'ON' if ckpt else 'OFF'} (BASELINE configs[4])")) if full_model
                                    else f"DEPTH-REDUCED AF3 ({args.enc_layers} enc + {args.dec_layers} dec layers) - not the BASELINE config"),
                       "micro_batch_per_gpu": args.batch, "global_batch": args.batch * world, "seq_len": s_tok, "audio_tokens": n_audio_tok,
                       "windows_per_sample": windows, "activation_checkpointing": ckpt,
                       "parallelism": f"dp{world}", "params": model.trainable_numel()},
            "loss": final_loss, "peak_mem_gib": round(peak_mem, 1), "host_enqueue_ms_per_step": round(1000.0 * host_enqueue / args.steps, 1),
            "waited_for_free_hbm_s": waited,
            "model_tflops_per_gpu": model_tf, "model_frac_of_mfma_peak": (model_tf / 2500.0) if model_tf else None,
            "hardware_tflops_per_gpu": hw_tf,
            "roofline": {"bound": "mfma", "kernel": "gemm_nt_bf16_k256 (+ k128 for <192-tile shapes): every dense contraction (fwd, dgrad, wgrad, conv stem, lm_head)",
                         "achieved": achieved_tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": achieved_tf / 2500.0, "traffic": traffic, "traffic_detail": traffic_detail,
                         "launches": gemm_launches, "avg_launch_ms": gemm_ms / max(gemm_launches, 1),
                         "gemm_ms_per_step": gemm_ms / prof_steps,
                         "timed_region_overlapped": {"achieved": (ov_flops / (ov_ms * 1e-3) / 1e12) if ov_ms > 0 else None, "launches": ov_launches,
                                                     "note": "same brackets inside the timed region; inflated when two streams share the chip"},
                         "note": ("HIP events around every GEMM launch on its launch stream; measured on one extra untimed step with the wgrad stream "
                                  "disabled (launches back to back)" if had_side else "HIP events around every GEMM launch, timed region")},
        }
        if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N = 1 only
            try:
                res["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the baseline must never take the measured line down
                res["cpu_baseline"] = {"value": None, "unit": "audio-s/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
        line = json.dumps(res)
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
        dist.destroy_process_group()
        drain()
    if line is not None:
        if world > 1:
            time.sleep(2.0)  # the other ranks have nothing left to print, let them leave: the JSON line stays the last line of output
        print(line, flush=True)


if __name__ == "__main__":
    main()
