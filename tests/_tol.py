"""Shared tolerances of the model-level parity tests (north_star: "token indices bit-exact, logits within a stated fp tolerance").

Reference = fp32 CPU on bf16-rounded weights; ours = bf16 storage / fp32 accumulate, logits emitted in bf16 (as the reference's own
bf16 run emits them).  A bf16 value carries 8 significant bits: at the logit scale of the trained goldens (|x|max ~ 19) one ulp is
0.125, so an absolute bar makes no sense across scales.  The bar is RELATIVE to the largest reference logit:
    |d logit| <= LOGIT_RTOL * max(1, |ref|max)            LOGIT_RTOL = 2^-6  (two bf16 ulps of the largest logit)
Token ids: argmax must equal the reference's wherever the reference's top-1/top-2 gap exceeds 2x the logit bar ("confident"
positions - >= 97 % of all valid positions on the trained goldens); generate() ids must be identical.

Two kinds of goldens (round 3):

  SMOOTH (cases D / E: random-init weights, logits O(1), a label on every text position).  The loss surface is smooth, bf16 rounding moves a
  gradient by about a percent: EVERY parameter gradient (67 tensors) is held to the FIXED bar GRAD_REL_L2 = 6e-2 - no noise-floor relaxation.

  SHARP (cases A / B / C: the TRAINED tiny reference; sharp softmax).  bf16 rounding at a handful of high-loss positions moves logits by
  several ulps and the loss gradient by tens of percent in ANY bf16 implementation, the reference's own included.  SURVEY.md §8c's rule is
  "ours <= 2 x the reference's own bf16 deviation": FLOOR_FACTOR = 2 (round 2 used 4 on ONE noisy sample).  The floor is now a DISTRIBUTION:
  the reference's bf16 run on this device against its fp32 run, over N_FLOOR_SEEDS fresh seeded batches of the same kind plus the stored
  golden batch (tests/test_model_gpu.py::_floor_distribution).  Every statistic s of ours must satisfy
        s(ours, golden batch)        <= max(absolute bar, min(FLOOR_FACTOR * max over batches of s(reference bf16),
                                                                  ABS_RELAX * absolute bar        [logits, loss]
                                                                  max(FLOOR_FACTOR * median, 1.25 * max) over batches [gradients]))   (round 4: see floor_bar)
        median over batches s(ours)  <= max(absolute bar, FLOOR_FACTOR * median over batches of s(reference bf16))
  Gradient bars are capped at GRAD_CAP; a tensor whose floor exceeds NOISE_DOMINATED on EVERY batch (the encoder's cancellation-dominated
  q_proj.bias: 5-8x its fp32 norm in the reference's own bf16 run) carries no information at bf16 on these goldens - it is reported, not
  asserted, and pinned by the smooth cases instead.  All figures go to profiles/r03_model_parity_report.json.
"""
LOGIT_RTOL = 2.0 ** -6
LOSS_ATOL = 1e-2
GRAD_REL_L2 = 6e-2     # per-tensor relative L2 error of a parameter gradient (bf16 gradient storage, eps 2^-8 per element)
AUDIO_REL_L2 = 2e-2
FLOOR_FACTOR = 2.0
GRAD_CAP = 0.5         # no sharp-golden gradient bar above this, whatever the floor
NOISE_DOMINATED = 0.5  # floor above this on every batch: bf16 cannot resolve the tensor on this golden (reported, pinned by the smooth cases)
N_FLOOR_SEEDS = 8


def logit_tol(ref_absmax: float) -> float:
    return LOGIT_RTOL * max(1.0, float(ref_absmax))


ABS_RELAX = 3.0        # round 4: a non-gradient statistic on the stored golden may exceed its absolute bar by at most this factor,
                       # however large the reference's own worst bf16 batch was (a bar of 2 x floor_max was up to 25 x what it measured)


def floor_bar(abs_bar: float, floor_values, cap=None, is_grad=False) -> float:
    """bar of a statistic OF THE STORED GOLDEN BATCH given the reference-bf16 values of the same statistic over the floor batches.
    The floor is heavy-tailed (reference logit-max 0.23 ... 8.4 across batches), so its MAXIMUM is no bar for one batch of ours:
      logits / loss:  max(abs, min(FLOOR_FACTOR x floor_max, ABS_RELAX x abs))
      gradients:      max(abs, min(FLOOR_FACTOR x floor_max, max(FLOOR_FACTOR x floor_median, 1.25 x floor_max), cap))"""
    fl = [float(v) for v in floor_values]
    relaxed = FLOOR_FACTOR * max(fl)
    # gradients: twice the typical reference batch, but never below what the reference's own worst batch reached plus a quarter (these rel-L2
    # figures are noise realisations of 0.2-0.5 on the sharp goldens: a kernel change that only re-orders fp32 sums moved ours 0.38 -> 0.49)
    relaxed = min(relaxed, max(FLOOR_FACTOR * median(fl), 1.25 * max(fl))) if is_grad else min(relaxed, ABS_RELAX * float(abs_bar))
    b = max(float(abs_bar), relaxed)
    return b if cap is None else min(float(cap), b)


def floor_range_bar(abs_bar: float, floor_values) -> float:
    """bar for a run of the REFERENCE's own bf16 model (tests/test_hf_plugin_gpu.py: the stock model with only its attention dispatched to libafk.so):
    both sides are the same noisy bf16 model, so the bar is the stock run's own range over the floor batches - max(abs, FLOOR_FACTOR x its maximum).
    (Our implementation's golden-batch statistics are held to the tighter floor_bar above.)"""
    return max(float(abs_bar), FLOOR_FACTOR * max(float(v) for v in floor_values))


def median(values):
    v = sorted(float(x) for x in values)
    n = len(v)
    return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])
