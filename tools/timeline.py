"""Timeline view of one training step from a rocprofv3 kernel trace (.db): GPU busy / idle, per-queue busy time, how much of
the step has >= 2 kernels in flight, and the per-kernel table of that step.   usage: python tools/timeline.py <results.db> [out.md]"""
import collections
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n)[:90]


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    rows = cur.execute("select name,start,end,queue_id from kernels order by start").fetchall()
    marks = [r[1] for r in rows if "logmel_kernel" in r[0]]  # first kernel of every step
    if len(marks) < 2:
        print("need >= 2 steps in the trace")
        return
    t0, t1 = marks[-2], marks[-1]
    step = [r for r in rows if t0 <= r[1] < t1]
    iv = sorted((r[1], r[2]) for r in step)
    ev = sorted([(s, 1) for s, _ in iv] + [(e, -1) for _, e in iv])
    c, last, one, multi = 0, None, 0, 0
    for t, d in ev:
        if last is not None:
            if c >= 2:
                multi += t - last
            elif c == 1:
                one += t - last
        c += d
        last = t
    byq = collections.defaultdict(float)
    agg = {}
    for n, s, e, q in step:
        byq[q] += (e - s) / 1e6
        a = agg.setdefault(short(n), [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e6
    out = [f"step (logmel -> next logmel): {(t1 - t0) / 1e6:.1f} ms, {len(step)} kernels",
           f"GPU busy (>= 1 kernel): {(one + multi) / 1e6:.1f} ms; idle {(t1 - t0 - one - multi) / 1e6:.1f} ms; exactly 1 kernel {one / 1e6:.1f} ms; >= 2 kernels {multi / 1e6:.1f} ms",
           "kernel time per HSA queue (ms): " + ", ".join(f"q{q}: {v:.1f}" for q, v in sorted(byq.items())),
           f"sum of kernel durations: {sum(byq.values()):.1f} ms", "",
           "| kernel | calls | total ms | avg us |", "|---|---|---|---|"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append(f"| {k} | {a[0]} | {a[1]:.2f} | {1e3 * a[1] / a[0]:.1f} |")
    txt = "\n".join(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")
    print("\n".join(out[:5]))


if __name__ == "__main__":
    main()
