"""HBM traffic of the single-sequence decode-chain launches from the PMC counters (one `rocprofv3 --pmc FETCH_SIZE` pass over tools/bench_decode_chain.py,
gfx950 correction of MI355X_MICROARCH.md: wide coalesced streams report half their bytes) against the algorithmic bytes (every weight once):
    python tools/measure_decode_traffic.py [out.json]        (on the GPU box; ~1 min)"""
import json, os, sqlite3, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, Hq, Hkv, D, I, V = 3584, 28, 4, 128, 18944, 152064
ALG = {"<1, 0,": ("qkv", 2.0 * (Hq + 2 * Hkv) * D * H), "<1, 2,": ("gate_up", 2.0 * 2 * I * H), "<1, 3,": ("lm_head", 2.0 * V * H),
       "<0, 1, 4": ("o_proj", 2.0 * H * Hq * D), "<0, 1, 8": ("down", 2.0 * H * I)}


def main():
    sys.path.insert(0, ROOT)
    from audio_flamingo_amd import _lib

    d = tempfile.mkdtemp(prefix="afk_pmc_", dir="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", "FETCH_SIZE", "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "tools", "bench_decode_chain.py")]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", ITERS="12"), capture_output=True, text=True, timeout=600)
    dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
    if not dbs:
        raise RuntimeError(f"rocprofv3 produced no database: rc {r.returncode}\n{r.stdout[-1500:]}\n{r.stderr[-1500:]}")
    cur = sqlite3.connect(dbs[0]).cursor()
    cols = [c[1] for c in cur.execute("pragma table_info(counters_collection)")]
    ix = {c: i for i, c in enumerate(cols)}
    name_col = ix.get("kernel_name", ix.get("name", 0))
    vals = {}
    for row in cur.execute("select * from counters_collection").fetchall():
        n = str(row[name_col])
        if "gemv_chain_kernel" in n and row[ix["counter_name"]] == "FETCH_SIZE":
            for key, (label, alg) in ALG.items():
                if "gemv_chain_kernel" + key in n:
                    vals.setdefault(label, []).append(float(row[ix["value"]]))
    out = {"afk_build_id": _lib.load().afk_build_id().decode(), "counter": "FETCH_SIZE (KB), one rocprofv3 --kernel-trace --pmc pass over tools/bench_decode_chain.py (ITERS=12)",
           "note": "hbm_read_bytes = 2 * FETCH_SIZE * 1024 (gfx950: wide coalesced streams report half their bytes, MI355X_MICROARCH.md HBM section)", "launches": {}}
    for key, (label, alg) in ALG.items():
        v = vals.get(label, [])
        if v:
            fetch = sum(v) / len(v)
            out["launches"][label] = {"n": len(v), "FETCH_SIZE_KB": fetch, "hbm_read_bytes": 2 * fetch * 1024, "algorithmic_weight_bytes": alg, "read_over_algorithmic": 2 * fetch * 1024 / alg}
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "decode_traffic.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
