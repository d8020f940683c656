"""Summarise a rocprofv3 sqlite (.db) kernel trace into a per-kernel table (the `--stats` view).
usage: python tools/rocpd_stats.py <results.db> [out.md] [--steps N]

--steps N (round 5): the trace holds N training steps; a group table is added in which a kernel counts as PER-STEP work when its
call count is a multiple of N (total / N) and as ONE-OFF work otherwise (model initialisation, parameter checksums, the bench's bookkeeping: reported
once, not per step) - round 4's "ATen / runtime 17 ms per step" had divided one-off launches by the step count."""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n[:100]


GROUPS = (("GEMM", r"gemm_|splitk"), ("AdamW", r"adamw"), ("attention", r"attn_|gqa_reduce"), ("SwiGLU backward", r"silu_mul_bwd"),
          ("norms (incl. gradient folds)", r"norm_|fold_partials"), ("transposes (W^T shadows, conv operands)", r"transpose|im2col|col2im|conv_weight"),
          ("bias column sums", r"colsum"), ("GELU backward", r"gelu_bwd"), ("RoPE", r"rope"), ("log-mel / CE / embed / other afk", r"logmel|ce_fwd|embed_|scale_add|loss_|avgpool|gather|scatter|count_valid|placeholder|sumsq|clip_coef|set_f32|silu_mul_fwd|gelu_fwd|cast_"))


def grouped(rows, steps, marker="logmel_kernel"):
    """per-step vs one-off work by TIME WINDOWS: a step is the interval between two launches of the marker kernel (the log-mel frontend opens every
    training step).  Per group: the MEDIAN over the complete interior intervals (the first may carry one-time allocations, the last runs into the bench's
    bookkeeping) = ms per step; one-off = the group's total minus steps x that median (model initialisation, checksums, the final report).
    (A first version classified by "call count divisible by the step count": 48 one-off checksum casts passed as per-step work, 1 537 transposes - 256 per
    step plus one at start-up - as one-off.)"""
    rows = sorted(rows, key=lambda r: r[1])
    marks = [r[1] for r in rows if marker in r[0]]
    if len(marks) < 3:
        return [f"", f"(no group table: fewer than three '{marker}' launches in the trace)"]
    group_of = lambda k: next((name for name, pat in GROUPS if re.search(pat, k)), "ATen / runtime (torch kernels, copies, fills)")
    inter = [dict() for _ in range(len(marks) - 1)]
    total = {}
    for name, st, en in rows:
        g = group_of(short(name))
        d = (en - st) / 1e3
        t = total.setdefault(g, [0, 0.0])
        t[0] += 1
        t[1] += d
        i = sum(1 for m in marks if m <= st) - 1
        if 0 <= i < len(inter):
            c = inter[i].setdefault(g, [0, 0.0])
            c[0] += 1
            c[1] += d
    use = inter[1:] if len(inter) > 2 else inter     # interior steps
    med = lambda v: sorted(v)[len(v) // 2]
    nsteps = len(marks)
    lines = ["", f"## Groups ({nsteps} steps in the trace; per step = median over the {len(use)} interior step windows, a window = one '{marker}' launch to the next)", "",
             "| group | launches / step | ms / step | one-off launches | one-off ms (whole trace: initialisation, checksums, report) |", "|---|---|---|---|---|"]
    tot = 0.0
    per = {g: (med([w.get(g, [0, 0.0])[0] for w in use]), med([w.get(g, [0, 0.0])[1] for w in use])) for g in total}
    for g in sorted(total, key=lambda g: -per[g][1]):
        n_, ms_ = per[g]
        tot += ms_
        lines.append(f"| {g} | {n_} | {ms_ / 1e3:.2f} | {max(total[g][0] - nsteps * n_, 0)} | {max(total[g][1] - nsteps * ms_, 0.0) / 1e3:.2f} |")
    lines.append(f"\nper-step kernel time (serial sum of exclusive durations): {tot / 1e3:.1f} ms")
    return lines


def main():
    steps = 0
    if "--steps" in sys.argv:
        i = sys.argv.index("--steps")
        steps = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {k} | {a[0]} | {a[1] / 1e3:.2f} | {a[1] / a[0]:.1f} | {a[2]:.1f} | {a[3]:.1f} | {100 * a[1] / total:.1f} |")
    lines.append(f"\ntotal kernel time {total / 1e3:.1f} ms over {len(rows)} dispatches; columns of kernels view: {cols}")
    if steps > 0:
        lines += grouped(rows, steps)
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
