"""PMC table of the LDS-staged attention kernels ("fresh PMC table ... MFMA-busy, clock, VGPR/AGPR, waves/SIMD per kernel").

Per shape two `rocprofv3 --kernel-trace --pmc` passes over tools/one_attn.py (separate runs, no other trace domain):
    pass 1: SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
    pass 2: SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
MFMA-busy = MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); clock = (GRBM_GUI_ACTIVE / 8) / duration (MI355X_MICROARCH.md, PMC section;
same reading as profiles/r02_attn_probes.md §4 and profiles/r03_gemm_pmc.md).  Algorithmic flops: forward 4 S^2 D H B (half when causal), dQ kernel 1.5x,
dK/dV sweep 2x of that (the two-kernel backward executes 3.5x for the algorithmic 2.5x).  Registers / waves per SIMD: `make report` (static).

    python tools/measure_attn_pmc.py [out.md] [shape ...]        (on the GPU box; ~25 s per shape)
"""
import os
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = {"dec": ("decoder B=8 S=1024 Hq=28 Hkv=4 D=128 causal", 8, 1024, 28, 128, True), "enc": ("encoder B=8 S=1500 H=20 D=64", 8, 1500, 20, 64, False),
          "long5": ("5-min decoder B=1 S=7774 Hq=28 Hkv=4 D=128 causal", 1, 7774, 28, 128, True),
          "long10": ("10-min decoder B=1 S=15274 Hq=28 Hkv=4 D=128 causal", 1, 15274, 28, 128, True)}
KERNELS = (("attn_fwd_lds_kernel", 1.0), ("attn_bwd_dq_lds_kernel", 1.5), ("attn_bwd_dkdv_lds_kernel", 2.0), ("gqa_reduce_kernel", 0.0))
MAX_GHZ = 2.4   # MI355X peak engine clock (MI355X_MICROARCH.md chip table)
P1 = ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]
P2 = ["SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS"]


def one_pass(counters, which, reps=3):
    d = tempfile.mkdtemp(prefix="afk_pmc_", dir="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", *counters, "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "tools", "one_attn.py"), which, str(reps)]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=600)
    dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
    if not dbs:
        raise RuntimeError("rocprofv3 produced no db: " + r.stderr[-2000:])
    cur = sqlite3.connect(dbs[0]).cursor()
    cols = [c[1] for c in cur.execute("pragma table_info(counters_collection)")]
    ix = {c: i for i, c in enumerate(cols)}
    name_col = ix.get("kernel_name", ix.get("name", 0))
    vals, durs = {}, {}
    for row in cur.execute("select * from counters_collection").fetchall():
        for k, _ in KERNELS:
            if k in str(row[name_col]):
                vals.setdefault(k, {}).setdefault(row[ix["counter_name"]], []).append(float(row[ix["value"]]))
    for n, s, e in cur.execute("select name, start, end from kernels").fetchall():
        for k, _ in KERNELS:
            if k in str(n):
                durs.setdefault(k, []).append((e - s) / 1e3)
    avg = lambda v: sum(v) / len(v)
    return {k: {c: avg(x) for c, x in d_.items()} for k, d_ in vals.items()}, {k: avg(v) for k, v in durs.items()}


def main():
    sys.path.insert(0, ROOT)
    from audio_flamingo_amd import _lib

    build = _lib.load().afk_build_id().decode()
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "attn_pmc.md")
    which = sys.argv[2:] or ["dec", "enc", "long5"]
    lines = [f"# PMC table of the LDS-staged attention kernels (build {build}, AFK_ATTN_WIDE={os.environ.get('AFK_ATTN_WIDE', '1')}, AFK_ATTN_DKDV={os.environ.get('AFK_ATTN_DKDV', 'default')})", "",
             "`python tools/measure_attn_pmc.py`: two `rocprofv3 --kernel-trace --pmc` passes per shape over `tools/one_attn.py` (3 forward + backward rounds, averages per launch;",
             "kernels alone on the chip, random bf16 operands).  cycles = min(GRBM_GUI_ACTIVE / 8, duration x 2.4 GHz): the GRBM window of a short kernel is wider than the kernel, so the",
             "clock column = cycles / duration is capped at the part's 2.4 GHz ('<=' where the cap applied) and MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles is then a lower bound;",
             "TF/s on ALGORITHMIC flops (causal = half the square; dQ 1.5x, dK/dV 2x the forward).  WAIT_ANY = parked at s_waitcnt / barrier, WAIT_INST_ANY = issue stalls,",
             "ACTIVE = issuing (fractions of SQ_WAVE_CYCLES); VALU, LDS = instructions per launch (chip sums).", "",
             "| shape | kernel | us (under PMC) | TF/s (algorithmic) | frac of 2.5 PF | **MFMA-busy** | clock GHz | WAIT_ANY | WAIT_INST_ANY | ACTIVE | WAIT_INST_LDS | VALU insts | LDS insts | LDS bank conflict cycles |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for w in which:
        label, B, S, H, D, causal = SHAPES[w]
        fwd = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
        a, us = one_pass(P1, w)
        b, _ = one_pass(P2, w)
        for k, mult in KERNELS:
            if k not in a or k not in us:
                continue
            # GRBM_GUI_ACTIVE counts from the dispatch's first activity to its last - for a kernel of tens of microseconds that window is wider than the
            # kernel's own duration, and cycles / duration came out ABOVE the part's 2.4 GHz maximum (round 4: 2.89 for the S = 1024 dQ kernel, 3.1-3.7
            # for gqa_reduce).  The cycles a kernel can have had are bounded by duration x 2.4 GHz: the clock column is capped there (marked "<="), and
            # MFMA-busy is taken against min(window cycles, duration x 2.4 GHz), i.e. it can only be UNDER-stated, never inflated by a short window
            cap = us[k] * MAX_GHZ * 1e3
            capped = a[k]["GRBM_GUI_ACTIVE"] / 8.0 > cap
            cyc = min(a[k]["GRBM_GUI_ACTIVE"] / 8.0, cap)
            busy = a[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc
            tf = fwd * mult / (us[k] * 1e-6) / 1e12
            bb = b.get(k, {})
            wc = bb.get("SQ_WAVE_CYCLES", float("nan"))
            f = lambda c: f"{bb.get(c, float('nan')) / wc:.2f}"
            lines.append(f"| {label} | `{k}<{D}>` | {us[k]:.1f} | {tf:.0f} | {tf / 2500:.3f} | **{busy:.2f}** | {'<= ' if capped else ''}{cyc / us[k] / 1e3:.2f} | {f('SQ_WAIT_ANY')} | {f('SQ_WAIT_INST_ANY')} | "
                         f"{f('SQ_ACTIVE_INST_ANY')} | {f('SQ_WAIT_INST_LDS')} | {bb.get('SQ_INSTS_VALU', float('nan')):.3g} | {bb.get('SQ_INSTS_LDS', float('nan')):.3g} | "
                         f"{bb.get('SQ_LDS_BANK_CONFLICT', float('nan')):.3g} |")
    out = "\n".join(lines) + "\n"
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    open(out_path, "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
