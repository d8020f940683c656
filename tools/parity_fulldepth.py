#!/usr/bin/env python
"""Full-depth parity record on the configuration the benchmark times (VERDICT r04 "missing" #2 / "next" #1).

BASELINE configs[1]: AF3-7B (32 encoder + 28 decoder layers), B = 8, S = 1024, one 30 s window per sample - ONE state_dict in
  (1) the live reference  transformers.AudioFlamingo3ForConditionalGeneration in fp32 on this GPU      = ground truth
      (modeling_audioflamingo3.py:584-642, modeling_qwen2.py:342-402; eager, sdpa, micro-batches of 2 with fp32 gradient accumulation),
  (2) the same reference in bf16 on this GPU (full batch, as bench.py's eager leg runs it)                = the noise floor of ANY bf16 implementation,
  (3) this repo's model (libafk.so)                                                                      = what is checked,
all three on the SAME synthetic batch (bench.synthetic_batch).  The reference gets its features from ITS OWN frontend
(WhisperFeatureExtractor on the host, feature_extraction_whisper.py:135-168); ours computes them with afk_logmel on the device - the comparison is end
to end from the waveform.

Compared: first-step loss; logits on the 2 048 labelled rows; argmax on the rows whose fp32 top-1/top-2 gap exceeds twice the logit bar; EVERY parameter
gradient (rel-L2 against fp32, beside the reference-bf16 figure for the same tensor); gradient norm per layer bucket; the first AdamW update (sign
agreement with the update the fp32 gradients imply, parameter-sum delta).  BASELINE configs[4] (one 5-minute clip: 10 windows, S = 7 774) is compared
forward-only (loss, labelled logits, argmax).

Bars (tests/_tol.py): |d loss| <= 1e-2; |d logit| <= 2^-6 * max(1, |ref|max) or 2 x the reference-bf16 figure; 0 argmax mismatches on confident rows;
per-tensor gradient rel-L2 <= max(6e-2, 2 x the reference-bf16 figure of that tensor) ("ours <= 2 x (oracle bf16 vs fp32)").

Used by bench.py (untimed leg -> `parity_fulldepth` in the JSON line) and tests/test_fullwidth_gpu.py.  Needs ~230 GiB of HBM at its peak (our replica
with fp32 optimizer state beside the kept fp32 reference gradients).  The reference is imported as the CHECKER only.
"""
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LOGIT_RTOL, LOSS_ATOL, GRAD_REL_L2, FLOOR_FACTOR, NOISE_DOMINATED = 2.0 ** -6, 1e-2, 6e-2, 2.0, 0.5   # tests/_tol.py
BF = torch.bfloat16


def _rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _sum64(t, chunk=1 << 28):
    """fp64 sum of a large tensor without a full-size temporary"""
    f = t.reshape(-1)
    return float(sum(f[i:i + chunk].double().sum() for i in range(0, f.numel(), chunk)))


def _free():
    gc.collect()
    torch.cuda.empty_cache()


def _group(key):
    """layer bucket of a reference parameter name (the arena's buckets: one per transformer layer + the stems and heads)"""
    p = key.split(".")
    if "layers" in p:
        i = p.index("layers")
        return ("enc" if "audio_tower" in key else "dec") + f".{int(p[i + 1]):02d}"
    if "audio_tower" in key:
        return "enc.stem+ln"
    if "multi_modal_projector" in key:
        return "projector"
    if key == "lm_head.weight":
        return "lm_head"
    return "dec.embed+norm"


def restore_rope_buffers(model):
    """`module.to(torch.bfloat16)` rounds EVERY floating buffer - including the rotary embedding's non-persistent `inv_freq` (modeling_qwen2.py:67-68),
    and a later `.float()` only upcasts the rounded values: at theta = 1e6 and position 1 000 the highest frequencies are then off by radians.  A
    checkpoint loaded with `from_pretrained(dtype=bf16)` never goes through `.to()` and keeps the fp32 buffer (`:87` computes it with an explicit fp32
    arange) - that is the behaviour this repo implements, so the harness puts the fp32 values back after every dtype change of the reference."""
    for mod in model.modules():
        if hasattr(mod, "inv_freq") and hasattr(mod, "compute_default_rope_parameters") and getattr(mod, "rope_type", "default") == "default":
            inv, _ = mod.compute_default_rope_parameters(mod.config, mod.inv_freq.device)
            mod.inv_freq = torch.nn.Buffer(inv.float(), persistent=False)
            mod.original_inv_freq = torch.nn.Buffer(inv.float().clone(), persistent=False)
            if hasattr(mod, "position_angles") and hasattr(mod, "_compute_position_angles"):   # Music Flamingo's rotary TIME embedding: its angle table and
                mod.position_angles = torch.nn.Buffer(mod._compute_position_angles(mod.inv_freq), persistent=False)   # every timestamp take inv_freq's dtype
    return model


def _reference_features(waves_np):
    """the reference's own frontend, on the host (feature_extraction_whisper.py:135-168; chunk length 30 s, 128 mel bins)"""
    from transformers import WhisperFeatureExtractor

    fe = WhisperFeatureExtractor(feature_size=128)
    out = fe([w for w in waves_np], sampling_rate=16000, return_tensors="pt", return_attention_mask=True, padding="max_length")
    return out["input_features"], out["attention_mask"]


def _labelled_rows(labels):
    """rows whose NEXT token carries a label (loss_utils.py:65-66 shift): [B, S] bool"""
    return torch.nn.functional.pad(labels, (0, 1), value=-100)[:, 1:] != -100


def _ref_forward(model, ids, feats, fmask, labels, dtype, micro, backward):
    """reference forward (+ backward with gradient accumulation in the parameter dtype's .grad) -> loss (python float), labelled-row logits [n, V] (dtype)"""
    B = ids.shape[0]
    wps = feats.shape[0] // B          # windows per sample (window-major inside a sample, bench.synthetic_batch)
    n_micro = (B + micro - 1) // micro
    sel_all = _labelled_rows(labels)
    losses, lg = [], []
    for i in range(0, B, micro):
        kw = dict(input_ids=ids[i:i + micro], input_features=feats[i * wps:(i + micro) * wps].to(dtype), input_features_mask=fmask[i * wps:(i + micro) * wps], labels=labels[i:i + micro])
        if backward:
            out = model(**kw)
            (out.loss / n_micro).backward()
        else:
            with torch.no_grad():
                out = model(**kw)
        losses.append(float(out.loss.detach()))
        lg.append(out.logits.detach()[sel_all[i:i + micro]].clone())
        del out
    # every sample carries the same number of labelled tokens (bench.synthetic_batch): the batch mean is the mean of the micro-batch means
    return float(np.mean(losses)), torch.cat(lg, 0)


def _logit_stats(got, ref32, noise=None):
    """noise: the reference-bf16 run's own max |d logit| on these rows (None while that run itself is being scored).  "Confident" rows (SURVEY.md §8c:
    "argmax equal wherever the fp32 top-1/top-2 gap exceeds the measured bf16 noise"): gap > 2 x max(logit bar, measured noise) - top-1 may fall and
    top-2 rise by the noise each.  The count under the bar alone (tests/_tol.py's rule at the tiny goldens' depth) is reported beside it."""
    d = (got.float() - ref32).abs()
    absmax = float(ref32.abs().max())
    tol = LOGIT_RTOL * max(1.0, absmax)
    top2 = ref32.topk(2, -1).values
    gap = top2[:, 0] - top2[:, 1]
    wrong = got.float().argmax(-1) != ref32.argmax(-1)
    conf_bar = gap > 2 * tol
    out = {"max_abs_err": float(d.max()), "rel_l2": _rel(got, ref32), "ref_absmax": absmax, "abs_bar": tol, "rows": int(ref32.shape[0]),
           "confident_rows_bar_only": int(conf_bar.sum()), "argmax_mismatches_bar_only": int((wrong & conf_bar).sum()),
           "argmax_agreement_all_rows": float((~wrong).float().mean())}
    if noise is not None:
        conf = gap > 2 * max(tol, noise)
        out.update({"confident_gap": 2 * max(tol, noise), "confident_rows": int(conf.sum()), "argmax_mismatches_on_confident_rows": int((wrong & conf).sum())})
    return out


def run(dev, batch=8, micro_fp32=2, long_windows=10, lr=1e-5, enc_layers=32, dec_layers=28, log=lambda s: print(s, file=sys.stderr, flush=True)):
    import bench
    from transformers import AudioFlamingo3ForConditionalGeneration as Ref

    t_start = time.perf_counter()
    cfg = bench.af3_7b_config(enc_layers, dec_layers)
    waves, ids, labels = bench.synthetic_batch(batch, 0, dev, 1)
    lwaves, lids, llabels = bench.synthetic_batch(1, 0, dev, long_windows) if long_windows else (None, None, None)
    feats_ref, fmask = (t.to(dev) for t in _reference_features(waves.cpu().numpy()))
    if long_windows:
        lfeats_ref, lfmask = (t.to(dev) for t in _reference_features(lwaves.cpu().numpy()))

    # ---------------------------------------------------------------- one state_dict: the reference's own init, every trivial tensor perturbed
    torch.manual_seed(0)
    with torch.device(dev):
        ref = Ref(cfg)
    g = torch.Generator(device=dev).manual_seed(4)
    with torch.no_grad():
        for k, p in ref.named_parameters():   # biases (zero) and norm weights (one) off their trivial init so that every gradient path carries signal
            if k.endswith(".bias"):
                p.copy_(0.02 * torch.randn(p.shape, device=dev, generator=g))
            elif "norm" in k.split(".")[-2] and k.endswith(".weight"):
                p.copy_(1 + 0.05 * torch.randn(p.shape, device=dev, generator=g))
    ref.to(BF)
    sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    ref.train()
    res = {"config": ("BASELINE configs[1]: " if (enc_layers, dec_layers, batch) == (32, 28, 8) else "NOT the BASELINE config: ") +
                     "AF3-7B %d+%d layers, B=%d, S=%d, one 30 s window per sample, bf16" % (enc_layers, dec_layers, batch, ids.shape[1]),
           "weights": "reference _init_weights under torch.manual_seed(0), biases ~ N(0, 0.02), norm weights ~ 1 + N(0, 0.05), rounded to bf16; "
                      "the SAME state_dict in all three models",
           "truth": f"reference in fp32 on this GPU (bf16-rounded weights upcast), micro-batches of {micro_fp32}, fp32 gradient accumulation",
           "floor": "reference in bf16 on this GPU (eager, sdpa), full batch", "bars": {"loss_abs": LOSS_ATOL, "logit_rtol": LOGIT_RTOL, "grad_rel_l2": GRAD_REL_L2,
                                                                                      "floor_factor": FLOOR_FACTOR}}

    # ---------------------------------------------------------------- (1) truth: fp32
    restore_rope_buffers(ref.float())
    t0 = time.perf_counter()
    loss32, lg32 = _ref_forward(ref, ids, feats_ref, fmask, labels, torch.float32, micro_fp32, True)
    g32 = {k: p.grad.detach().clone() for k, p in ref.named_parameters() if p.grad is not None}
    ref.zero_grad(set_to_none=True)
    long32 = None
    if long_windows:
        long32 = _ref_forward(ref, lids, lfeats_ref, lfmask, llabels, torch.float32, 1, False)
    torch.cuda.synchronize()
    log(f"[parity] fp32 reference: loss {loss32:.6f}, {len(g32)} gradient tensors, {time.perf_counter() - t0:.1f} s")

    # ---------------------------------------------------------------- (2) floor: the reference's own bf16 run
    restore_rope_buffers(ref.to(BF))
    assert all(torch.equal(v, sd[k]) for k, v in list(ref.state_dict().items())[:8]), "bf16 -> fp32 -> bf16 must give the weights back"
    t0 = time.perf_counter()
    loss16, lg16 = _ref_forward(ref, ids, feats_ref, fmask, labels, BF, batch, True)
    floor_g = {k: _rel(p.grad, g32[k]) for k, p in ref.named_parameters() if p.grad is not None}
    floor_sign = {}
    gn16 = {}
    for k, p in ref.named_parameters():
        if p.grad is not None:
            gn16[_group(k)] = gn16.get(_group(k), 0.0) + float(p.grad.float().square().sum())
            floor_sign[k] = (int((torch.sign(p.grad.float()) == torch.sign(g32[k])).sum()), p.numel())
    ref.zero_grad(set_to_none=True)
    long16 = None
    if long_windows:
        long16 = _ref_forward(ref, lids, lfeats_ref, lfmask, llabels, BF, 1, False)
    torch.cuda.synchronize()
    log(f"[parity] bf16 reference: loss {loss16:.6f}, worst gradient rel-L2 vs fp32 {max(floor_g.values()):.4f}, {time.perf_counter() - t0:.1f} s")
    floor_logits = _logit_stats(lg16, lg32)
    del ref, lg16
    _free()

    # ---------------------------------------------------------------- (3) ours
    from audio_flamingo_amd.arena import FusedAdamW
    from audio_flamingo_amd.frontend import LogMelFrontend
    from audio_flamingo_amd.modeling import AudioFlamingo3ForConditionalGeneration as Mine

    t0 = time.perf_counter()
    m = Mine(cfg, device=dev, init_seed=0)
    m.load_state_dict(sd)
    del sd
    opt = FusedAdamW(m.arena, lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    frontend = LogMelFrontend(dev)
    res["logmel_max_abs_diff_vs_reference_frontend"] = float((frontend(waves, out_dtype=torch.float32) - feats_ref).abs().max())
    if long_windows:   # forward only, BEFORE the optimizer moves the weights
        with torch.no_grad():
            lout = m(input_ids=lids, input_features=frontend(lwaves, out_dtype=BF), labels=llabels, return_logits=True)
        lsel = _labelled_rows(llabels)
        lloss, llg = float(lout.loss), lout.logits[lsel].clone()
        del lout
        lf = _logit_stats(long16[1], long32[1])
        ls = _logit_stats(llg, long32[1], lf["max_abs_err"])
        res["long5min_forward"] = {"config": f"BASELINE configs[4] shape: one 5-minute clip = {long_windows} windows, S = {lids.shape[1]}, forward only",
                                   "loss": lloss, "loss_ref_fp32": long32[0], "loss_ref_bf16": long16[0], "logits": ls, "logits_floor_ref_bf16": lf}
        del llg, long32, long16
        _free()
    m.zero_grad()
    out = m(input_ids=ids, input_features=frontend(waves, out_dtype=BF), labels=labels)
    out.loss.backward()
    m.arena.join_streams()
    torch.cuda.synchronize()
    loss = float(out.loss.detach())
    lg = out.logits[_labelled_rows(labels)].clone()
    del out
    res.update({"loss": loss, "loss_ref_fp32": loss32, "loss_ref_bf16": loss16, "loss_abs_err": abs(loss - loss32), "loss_abs_err_floor": abs(loss16 - loss32),
                "logits": _logit_stats(lg, lg32, floor_logits["max_abs_err"]), "logits_floor_ref_bf16": floor_logits})
    del lg, lg32
    params = dict(m.named_parameters())
    grads, bad, noisy, gn, gn32 = {}, {}, {}, {}, {}
    sign_ok = sign_n = fsign_ok = fsign_n = 0
    for k, gr in g32.items():
        p = params[k]
        if not p.requires_grad:
            continue
        assert p.grad is not None, k
        r = _rel(p.grad, gr)
        grads[k] = r
        if floor_g[k] > NOISE_DOMINATED:   # the reference's own bf16 run is off by more than half this tensor's fp32 norm: bf16 cannot resolve it on this
            noisy[k] = {"ours": r, "floor": floor_g[k], "fp32_norm": float(gr.norm())}   # batch (tests/_tol.py NOISE_DOMINATED) - reported, held to the floor itself
            if r > FLOOR_FACTOR * floor_g[k]:
                bad[k] = {"ours": r, "floor": floor_g[k]}
        elif r > max(GRAD_REL_L2, FLOOR_FACTOR * floor_g[k]):
            bad[k] = {"ours": r, "floor": floor_g[k]}
        gn[_group(k)] = gn.get(_group(k), 0.0) + float(p.grad.float().square().sum())
        gn32[_group(k)] = gn32.get(_group(k), 0.0) + float(gr.square().sum())
        sign_ok += int((torch.sign(p.grad.float()) == torch.sign(gr)).sum())
        sign_n += p.numel()
        fsign_ok += floor_sign[k][0]
        fsign_n += floor_sign[k][1]
    resolved = {k: v for k, v in grads.items() if k not in noisy}
    worst = max(resolved, key=resolved.get)
    # >= 10 tensors spread over depth, by name (the record lists every tensor's figure in `all_tensors`)
    el, dl = enc_layers - 1, dec_layers - 1
    spread = ["model.audio_tower.conv1.weight", "model.audio_tower.layers.0.self_attn.q_proj.weight", f"model.audio_tower.layers.{el // 2}.fc1.weight",
              f"model.audio_tower.layers.{el}.fc2.weight", "model.multi_modal_projector.linear_1.weight", "model.language_model.embed_tokens.weight",
              "model.language_model.layers.0.self_attn.q_proj.weight", "model.language_model.layers.0.mlp.gate_proj.weight",
              f"model.language_model.layers.{dl // 2}.self_attn.k_proj.bias", f"model.language_model.layers.{dl // 2}.mlp.down_proj.weight",
              f"model.language_model.layers.{dl}.mlp.up_proj.weight", f"model.language_model.layers.{dl}.post_attention_layernorm.weight",
              "model.language_model.norm.weight", "lm_head.weight"]
    ratios = sorted(grads[k] / max(floor_g[k], 1e-12) for k in grads)
    res["gradients"] = {"tensors": len(grads), "worst": {"name": worst, "ours": grads[worst], "floor": floor_g[worst]},
                        "median_rel_l2": float(np.median(list(grads.values()))), "median_floor": float(np.median([floor_g[k] for k in grads])),
                        "ours_over_floor": {"median": ratios[len(ratios) // 2], "p95": ratios[int(0.95 * len(ratios))], "max": ratios[-1]},
                        "over_bar": bad, "noise_dominated_in_the_reference_bf16_run": noisy, "spread": {k: {"ours": grads[k], "floor": floor_g[k]} for k in spread if k in grads},
                        "sign_agreement_with_fp32": sign_ok / max(sign_n, 1), "sign_agreement_floor": fsign_ok / max(fsign_n, 1),
                        "bucket_norms": {b: {"ours": gn[b] ** 0.5, "fp32": gn32[b] ** 0.5, "ref_bf16": gn16[b] ** 0.5} for b in sorted(gn)},
                        "bucket_norm_rel_err_max": max(abs(gn[b] ** 0.5 - gn32[b] ** 0.5) / max(gn32[b] ** 0.5, 1e-30) for b in gn),
                        "all_tensors": {k: [round(grads[k], 5), round(floor_g[k], 5)] for k in grads}}

    # ---------------------------------------------------------------- first AdamW step: the update against what the fp32 gradients imply
    # step 1 of AdamW (bias-corrected m = g, v = g^2, no decay): delta = -lr * g / (|g| + eps)
    base = m.arena.params.storage_offset()
    before = opt.master.clone() if hasattr(opt, "master") else None
    sum_before = _sum64(m.arena.params)
    opt.step()
    torch.cuda.synchronize()
    upd_ok = upd_n = 0
    d_sum = d_sum_exp = 0.0
    rel_num = rel_den = 0.0
    for k, gr in g32.items():
        p = params[k]
        if not p.requires_grad or not p.is_contiguous():
            continue
        off = p.storage_offset() - base
        d = (opt.master[off: off + p.numel()] - before[off: off + p.numel()]).view(p.shape)
        exp = -lr * gr / (gr.abs() + 1e-8)
        upd_ok += int((torch.sign(d) == torch.sign(exp)).sum())
        upd_n += p.numel()
        d_sum += float(d.double().sum())
        d_sum_exp += float(exp.double().sum())
        rel_num += float((d - exp).double().square().sum())
        rel_den += float(exp.double().square().sum())
    res["adamw_first_step"] = {"lr": lr, "update_sign_agreement_with_fp32_gradients": upd_ok / max(upd_n, 1),
                               "reference_bf16_gradient_sign_agreement_with_fp32": fsign_ok / max(fsign_n, 1),
                               "update_rel_l2_vs_fp32_implied": (rel_num / max(rel_den, 1e-300)) ** 0.5,
                               "master_sum_delta": d_sum, "master_sum_delta_fp32_implied": d_sum_exp,
                               "bf16_param_sum_before": sum_before, "bf16_param_sum_after": _sum64(m.arena.params),
                               "note": "|delta| = lr for every element with |g| >> eps, so the update differs from the fp32-implied one exactly where a gradient "
                                       "SIGN differs (near-zero gradients): sign agreement is the meaningful figure, beside the reference-bf16 gradients' own"}
    log(f"[parity] ours: loss {loss:.6f} (fp32 {loss32:.6f}, ref bf16 {loss16:.6f}); worst gradient {worst} {grads[worst]:.4f} (floor {floor_g[worst]:.4f}); "
        f"{time.perf_counter() - t0:.1f} s")
    del m, opt, before, g32, params
    _free()

    lgs = res["logits"]
    ok = {"loss": res["loss_abs_err"] <= max(LOSS_ATOL, FLOOR_FACTOR * res["loss_abs_err_floor"]),
          "logits": lgs["max_abs_err"] <= max(lgs["abs_bar"], FLOOR_FACTOR * floor_logits["max_abs_err"]),
          "argmax": lgs["argmax_mismatches_on_confident_rows"] == 0,
          "gradients": not bad,
          "bucket_norms": res["gradients"]["bucket_norm_rel_err_max"] <= GRAD_REL_L2,
          "adamw_update": res["adamw_first_step"]["update_sign_agreement_with_fp32_gradients"] >= res["adamw_first_step"]["reference_bf16_gradient_sign_agreement_with_fp32"] - 0.02}
    if long_windows:
        l5 = res["long5min_forward"]
        ok["long5min_loss"] = abs(l5["loss"] - l5["loss_ref_fp32"]) <= max(LOSS_ATOL, FLOOR_FACTOR * abs(l5["loss_ref_bf16"] - l5["loss_ref_fp32"]))
        ok["long5min_logits"] = l5["logits"]["max_abs_err"] <= max(l5["logits"]["abs_bar"], FLOOR_FACTOR * l5["logits_floor_ref_bf16"]["max_abs_err"])
        ok["long5min_argmax"] = l5["logits"]["argmax_mismatches_on_confident_rows"] == 0
    res["checks"] = ok
    res["green"] = all(ok.values())
    res["seconds"] = round(time.perf_counter() - t_start, 1)
    return res


def write_record(res, name="parity_fulldepth.json"):
    """the full record (every tensor) beside the bench line: gpurun_out/ (merged back by gpurun); best effort"""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name), "w") as f:
            json.dump(res, f, indent=1)
    except OSError:
        pass


def summary(res):
    """the part of the record that rides in bench.py's JSON line (the full record - every tensor - goes to gpurun_out/ and profiles/)"""
    g = res["gradients"]
    out = {k: res[k] for k in ("config", "weights", "truth", "floor", "bars", "loss", "loss_ref_fp32", "loss_ref_bf16", "loss_abs_err", "logits",
                               "logits_floor_ref_bf16", "logmel_max_abs_diff_vs_reference_frontend", "adamw_first_step", "checks", "green", "seconds") if k in res}
    out["gradients"] = {k: g[k] for k in ("tensors", "worst", "median_rel_l2", "median_floor", "ours_over_floor", "over_bar", "noise_dominated_in_the_reference_bf16_run", "spread", "sign_agreement_with_fp32",
                                          "sign_agreement_floor", "bucket_norm_rel_err_max")}
    if "long5min_forward" in res:
        out["long5min_forward"] = res["long5min_forward"]
    return out


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_fulldepth.json"))
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--no-long", action="store_true")
    ap.add_argument("--enc-layers", type=int, default=32)
    ap.add_argument("--dec-layers", type=int, default=28)
    a = ap.parse_args()
    r = run(torch.device("cuda", 0), batch=a.batch, long_windows=0 if a.no_long else 10, enc_layers=a.enc_layers, dec_layers=a.dec_layers)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(r, f, indent=1)
    print(json.dumps(summary(r)))
    sys.exit(0 if r["green"] else 1)
