"""MFMA utilisation of the 256x256 GEMM kernels from the PMC counters (rocprofv3 --pmc, --kernel-trace only, one pass per counter set):
    MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs)
(SQ_* counters are summed over the chip, GRBM_GUI_ACTIVE over the 8 XCDs: MI355X_MICROARCH.md, PMC section; same reading as profiles/r02_gemm_pmc.md).
The effective shader clock of the profiled launches = (GRBM_GUI_ACTIVE / 8) / kernel duration.  Output: markdown table, stamped with afk_build_id.

    python tools/measure_mfma_util.py [out.md]        (on the GPU box; ~2 min)
"""
import os
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("gate|up forward (NT)", "NT", 8192, 37888, 3584, "gemm_nt_bf16_k256"), ("gate|up weight gradient (TN)", "TN", 37888, 3584, 8192, "gemm_xt_bf16_k256"),
         ("down-proj forward (NT, 1.75 rounds)", "NT", 8192, 3584, 18944, "gemm_nt_bf16_k256"), ("encoder fc1 forward (NT, K = 1280)", "NT", 12000, 5120, 1280, "gemm_nt_bf16_k256")]


def one_pass(counters, form, M, N, K, kernel, reps=4):
    d = tempfile.mkdtemp(prefix="afk_pmc_", dir="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", *counters, "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "tools", "one_gemm.py"),
           str(M), str(N), str(K), "0", str(reps), form]
    subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=300)
    dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
    cur = sqlite3.connect(dbs[0]).cursor()
    cols = [c[1] for c in cur.execute("pragma table_info(counters_collection)")]
    ix = {c: i for i, c in enumerate(cols)}
    name_col = ix.get("kernel_name", ix.get("name", 0))
    vals = {}
    for row in cur.execute("select * from counters_collection").fetchall():
        if kernel in str(row[name_col]):
            vals.setdefault(row[ix["counter_name"]], []).append(float(row[ix["value"]]))
    dur = [(e - s) / 1e3 for n, s, e in cur.execute("select name, start, end from kernels").fetchall() if kernel in str(n)]
    return {k: sum(v) / len(v) for k, v in vals.items()}, (sum(dur) / len(dur) if dur else float("nan"))


def main():
    sys.path.insert(0, ROOT)
    from audio_flamingo_amd import _lib

    build = _lib.load().afk_build_id().decode()
    lines = [f"# MFMA utilisation of the GEMM kernels from PMC counters (build {build})", "",
             "`python tools/measure_mfma_util.py`: per shape two `rocprofv3 --kernel-trace --pmc` passes (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE | SQ_WAVE_CYCLES SQ_WAIT_INST_ANY",
             "SQ_ACTIVE_INST_ANY) over `tools/one_gemm.py` (4 launches averaged, uniform random bf16 operands).  MFMA-busy = MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8);",
             "clock = (GRBM_GUI_ACTIVE / 8) / duration; achieved = 2MNK / duration; frac of the 2.5 PF nominal peak = MFMA-busy x clock / 2.4 GHz up to rounding.", "",
             "| shape | kernel | us (under PMC) | TFLOP/s | frac of 2.5 PF | **MFMA-busy** | clock GHz | WAIT_INST_ANY / WAVE_CYCLES | ACTIVE_INST_ANY / WAVE_CYCLES |", "|---|---|---|---|---|---|---|---|---|"]
    for name, form, M, N, K, kernel in CASES:
        a, us = one_pass(["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"], form, M, N, K, kernel)
        b, _ = one_pass(["SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"], form, M, N, K, kernel)
        cyc = a["GRBM_GUI_ACTIVE"] / 8.0
        busy = a["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc
        tf = 2.0 * M * N * K / (us * 1e-6) / 1e12
        lines.append(f"| {name} {M} x {N} x {K} | `{kernel}` | {us:.0f} | {tf:.0f} | {tf / 2500:.2f} | **{busy:.2f}** | {cyc / us / 1e3:.2f} | "
                     f"{b['SQ_WAIT_INST_ANY'] / b['SQ_WAVE_CYCLES']:.2f} | {b['SQ_ACTIVE_INST_ANY'] / b['SQ_WAVE_CYCLES']:.2f} |")
    out = "\n".join(lines) + "\n"
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "mfma_util.md")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    open(path, "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
