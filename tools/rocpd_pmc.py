"""Per-kernel PMC averages from a rocprofv3 sqlite db: python tools/rocpd_pmc.py <db> [kernel-substring]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
rows = cur.execute("select * from counters_collection").fetchall()
ix = {c: i for i, c in enumerate(cols)}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r[ix.get("kernel_name", ix.get("name", 0))]
    if sub in str(name):
        agg[str(name)[:60]][r[ix["counter_name"]]].append(r[ix["value"]])
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
if not agg:
    print("columns:", cols); print(rows[:3])
