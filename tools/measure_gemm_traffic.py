"""HBM traffic of ONE launch of the dominant kernel (gemm_nt_bf16_k256 on the decoder gate|up shape 8192 x 37888 x 3584) from the PMC counters,
exactly as /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes: separate `--pmc` passes (FETCH_SIZE costs 3 of the 4
TCC slots, WRITE_SIZE 2), each with --kernel-trace only; FETCH_SIZE doubled (gfx950 tallies the 128-byte requests of wide coalesced reads at
64 bytes), WRITE_SIZE as reported.  The result is STAMPED with afk_build_id of the library that ran, and bench.py refuses to quote a traffic
figure whose stamp differs from the library it is timing (the quoted figure must not go stale silently).

    python tools/measure_gemm_traffic.py [out.json]          (on the GPU box; ~1 min)
"""
import json
import os
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPE = dict(M=8192, N=37888, K=3584)
KERNEL = "gemm_nt_bf16_k256"
PER_LAUNCH = {}


def one_pass(counters, reps=8):
    d = tempfile.mkdtemp(prefix="afk_pmc_", dir="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", *counters, "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "tools", "one_gemm.py"),
           str(SHAPE["M"]), str(SHAPE["N"]), str(SHAPE["K"]), "0", str(reps), "NT"]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=300)
    dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
    if not dbs:
        raise RuntimeError(f"rocprofv3 produced no database: rc {r.returncode}\n{r.stdout[-1500:]}\n{r.stderr[-1500:]}")
    cur = sqlite3.connect(dbs[0]).cursor()
    cols = [c[1] for c in cur.execute("pragma table_info(counters_collection)")]
    ix = {c: i for i, c in enumerate(cols)}
    name_col = ix.get("kernel_name", ix.get("name", 0))
    vals = {}
    for row in cur.execute("select * from counters_collection").fetchall():
        if KERNEL in str(row[name_col]):
            vals.setdefault(row[ix["counter_name"]], []).append(float(row[ix["value"]]))
    PER_LAUNCH.update({k: list(v) for k, v in vals.items()})   # round 5: every launch, not only the mean (the spread of the figure is the question)
    return {k: sum(v) / len(v) for k, v in vals.items()}, {k: len(v) for k, v in vals.items()}


def main():
    sys.path.insert(0, ROOT)
    from audio_flamingo_amd import _lib

    build = _lib.load().afk_build_id().decode()
    pmc, n = {}, {}
    for counters in (["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_HIT_sum", "TCC_MISS_sum"]):
        v, c = one_pass(counters)
        pmc.update(v), n.update(c)
    M, N, K = SHAPE["M"], SHAPE["N"], SHAPE["K"]
    alg = 2.0 * (M * K + N * K + M * N)
    hbm = 2.0 * pmc["FETCH_SIZE"] * 1024 + pmc["WRITE_SIZE"] * 1024
    out = {"kernel": KERNEL + "<0>", "afk_build_id": build, "shape": SHAPE,
           "pmc": {"FETCH_SIZE_KB": pmc["FETCH_SIZE"], "WRITE_SIZE_KB": pmc["WRITE_SIZE"], "TCC_HIT_sum": pmc.get("TCC_HIT_sum"), "TCC_MISS_sum": pmc.get("TCC_MISS_sum"),
                   "launches_averaged": n, "passes": "three separate runs: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum -- python tools/one_gemm.py 8192 37888 3584 0 8 NT"},
           "hbm_bytes_per_launch": hbm,
           "hbm_bytes_note": "2*FETCH_SIZE*1024 + WRITE_SIZE*1024: FETCH_SIZE reports 1/2 of the bytes of wide coalesced streams on gfx950 (MI355X_MICROARCH.md HBM section); WRITE_SIZE uncorrected",
           "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": hbm / alg,
           "l2_hit_rate": (pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"])) if pmc.get("TCC_HIT_sum") else None,
           # the same launch, the same operands, back to back in one process: how much of the run-to-run spread of this figure (4.0-4.7 GB over rounds 2-4)
           # is already there between consecutive launches
           "per_launch_read_gb": [round(2.0 * v * 1024 / 1e9, 3) for v in PER_LAUNCH.get("FETCH_SIZE", [])],
           "per_launch_write_gb": [round(v * 1024 / 1e9, 3) for v in PER_LAUNCH.get("WRITE_SIZE", [])]}
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "gemm_traffic.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
