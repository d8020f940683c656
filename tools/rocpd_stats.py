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


def grouped(agg, steps):
    per, once = {}, {}
    for k, a in agg.items():
        g = next((name for name, pat in GROUPS if re.search(pat, k)), None)
        if g is None:
            g = "ATen / runtime (torch kernels, copies, fills)"
        tgt = per if (a[0] >= steps and a[0] % steps == 0) else once
        t = tgt.setdefault(g, [0, 0.0])
        t[0] += a[0]
        t[1] += a[1]
    lines = [f"", f"## Groups ({steps} steps in the trace)", "", "| group | per-step launches | ms / step | one-off launches (not a multiple of the step count) | one-off ms (whole trace) |", "|---|---|---|---|---|"]
    tot = 0.0
    for g in sorted(set(per) | set(once), key=lambda g: -(per.get(g, [0, 0.0])[1])):
        p_, o_ = per.get(g, [0, 0.0]), once.get(g, [0, 0.0])
        tot += p_[1] / steps
        lines.append(f"| {g} | {p_[0] // steps} | {p_[1] / steps / 1e3:.2f} | {o_[0]} | {o_[1] / 1e3:.2f} |")
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
        lines += grouped(agg, steps)
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
