"""A/B of the round-6 attention forward / dQ schedule (afk_attn_set_sched: 1 = explicit read rings + interleaved LDS-DMA, 0 = the compiler-scheduled reads of
rounds 1-5): bit-equality of O, LSE and dQKV on every shape, then HIP-event timing of both schedules alternating (forward alone; backward = dQ + dK/dV (+ reduce)).
Per-kernel durations come from tools/runs/attn_sched_prof.sh (rocprofv3 kernel trace of tools/one_attn.py under AFK_ATTN_SCHED=0 / 1)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_flamingo_amd import ops, _lib
dev = torch.device("cuda")
SHAPES = [("decoder S=1024", 8, 1024, 28, 4, 128, True), ("encoder S=1500 D=64", 8, 1500, 20, 20, 64, False), ("decoder S=2048", 4, 2048, 28, 4, 128, True),
          ("5-min decoder S=7774", 1, 7774, 28, 4, 128, True), ("ragged S=1000 causal", 2, 1000, 28, 4, 128, True), ("small", 1, 128, 4, 2, 128, True),
          ("encoder-like S=777 D=64", 2, 777, 20, 20, 64, False)]
only = sys.argv[1:]
for name, B, S, Hq, Hkv, D, causal in SHAPES:
    if only and not any(o in name for o in only):
        continue
    qkv = (torch.randn((B * S, (Hq + 2 * Hkv) * D), device=dev) * 0.5).to(torch.bfloat16)
    do = (torch.randn((B * S, Hq * D), device=dev) * 0.5).to(torch.bfloat16)
    res = {}
    VAR = [(0, 0), (1, 0), (1, 1)]   # (schedule, XCD-aware block map)
    for sch in VAR:
        _lib.call("afk_attn_set_sched", sch[0])
        _lib.call("afk_attn_set_xcd_map", sch[1])
        o, lse = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
        dqkv = ops.attn_bwd(qkv, o, do, lse, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
        torch.cuda.synchronize()
        res[sch] = (o.clone(), lse[..., :S].clone(), dqkv.clone())
    row = {"shape": name, "bit_equal": {k: all(bool(torch.equal(res[VAR[0]][i], res[v][i])) for v in VAR[1:]) for i, k in enumerate(("O", "LSE", "dQKV"))}}
    ts = {(s_, w): [] for s_ in VAR for w in ("fwd", "bwd")}
    n = 20 if S < 4000 else 6
    for rnd in range(3):
        for sch in VAR:
            _lib.call("afk_attn_set_sched", sch[0])
            _lib.call("afk_attn_set_xcd_map", sch[1])
            for _ in range(2):
                o, lse = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
                ops.attn_bwd(qkv, o, do, lse, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            for _ in range(n):
                o, lse = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
            e[1].record()
            for _ in range(n):
                ops.attn_bwd(qkv, o, do, lse, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
            e[2].record()
            torch.cuda.synchronize()
            ts[(sch, "fwd")].append(round(1000 * e[0].elapsed_time(e[1]) / n, 1))
            ts[(sch, "bwd")].append(round(1000 * e[1].elapsed_time(e[2]) / n, 1))
    row["fwd_us"] = {f"sched{v[0]}_xcd{v[1]}": ts[(v, "fwd")] for v in VAR}
    row["bwd_us"] = {f"sched{v[0]}_xcd{v[1]}": ts[(v, "bwd")] for v in VAR}
    print(json.dumps(row), flush=True)
_lib.call("afk_attn_set_sched", 1)
_lib.call("afk_attn_set_xcd_map", 1)
