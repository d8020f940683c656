"""Shared tolerances of the model-level parity tests (north_star: "token indices bit-exact, logits within a stated fp tolerance").

Reference = fp32 CPU on bf16-rounded weights; ours = bf16 storage / fp32 accumulate, logits emitted in bf16 (as the reference's own
bf16 run emits them).  A bf16 value carries 8 significant bits: at the logit scale of the trained goldens (|x|max ~ 19) one ulp is
0.125, so an absolute bar makes no sense across scales.  The bar is RELATIVE to the largest reference logit:
    |d logit| <= LOGIT_RTOL * max(1, |ref|max)            LOGIT_RTOL = 2^-6  (two bf16 ulps of the largest logit)
and is cross-checked against the measured noise floor of the reference itself (its own bf16 eager-ROCm run vs its fp32 run on the same
golden, computed inside tests/test_model_gpu.py::test_forward_backward_vs_reference_golden).
Token ids: argmax must equal the reference's wherever the reference's top-1/top-2 gap exceeds 2x the logit bar ("confident"
positions - >= 97 % of all valid positions on the round-2 goldens); generate() ids must be identical.

Noise-floor rule (round 2).  The goldens come from a TRAINED tiny reference: its softmax is sharp, so bf16 rounding in a handful of
high-loss positions moves logits by several ulps and the loss gradient by tens of percent in ANY bf16 implementation - the reference's own
bf16 run included (gradient rel-L2 of 0.3-0.5 against its fp32 run on case A; cancellation-dominated tensors such as the encoder's
q_proj.bias are off by 5-7x their fp32 norm).  Each bar is therefore max(absolute bar, FLOOR_FACTOR x the deviation of the reference's bf16
run on this device from the same fp32 golden), FLOOR_FACTOR = 4; the logit RMS error, the robust statistic, must stay within 2x the floor.
The floor is ONE noisy sample of a heavy-tailed quantity (the same tensor's floor ranges from 0.05 to 1.2 across the three cases), hence 4
rather than the 2 of SURVEY.md §8c for the max-type statistics and the per-tensor gradient norms; the measured figures of both runs are written to profiles/r02_model_parity_report.json
(ours is at or below the floor on 30 of the 33 stored gradient tensors).
"""
LOGIT_RTOL = 2.0 ** -6
LOSS_ATOL = 1e-2
GRAD_REL_L2 = 6e-2     # per-tensor relative L2 error of a parameter gradient (bf16 gradient storage, eps 2^-8 per element)
AUDIO_REL_L2 = 2e-2
FLOOR_FACTOR = 4.0


def logit_tol(ref_absmax: float) -> float:
    return LOGIT_RTOL * max(1.0, float(ref_absmax))


def grad_bar(floor_rel: float) -> float:
    """allowed relative L2 deviation of a parameter gradient given the reference-bf16 deviation of the same tensor"""
    return max(GRAD_REL_L2, FLOOR_FACTOR * float(floor_rel))
