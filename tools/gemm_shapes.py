"""per-shape table of one serial training step from the per-launch CSV (AFK_PROF_DUMP=file python bench.py ...):
M,N,K,variant(1=nt128,2=nt256,3=nn256,4=tn256),ms -> launches, total ms, TFLOP/s, 256x256 tiles, rounds on 256 CUs"""
import sys, csv, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = (int(r["variant"]), int(r["M"]), int(r["N"]), int(r["K"]))
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += float(r["ms"])
name = {1: "nt128", 2: "nt256", 3: "nn256", 4: "tn256"}
tot = sum(a[1] for a in agg.values())
print("| kernel | M | N | K | launches | total ms | % | TFLOP/s | tiles | rounds |")
print("|---|---|---|---|---|---|---|---|---|---|")
for (v, M, N, K), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"| {name[v]} | {M} | {N} | {K} | {n} | {ms:.2f} | {100 * ms / tot:.1f} | {2.0 * M * N * K * n / ms / 1e9:.0f} | {tiles} | {tiles / 256:.2f} |")
print(f"\ntotal {tot:.1f} ms over {sum(a[0] for a in agg.values())} launches")
