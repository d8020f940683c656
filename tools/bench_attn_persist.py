"""A/B of the persistent forward attention (afk_attn2_fwd_persistent, AFK_ATTN_PERSIST) against the grid form: bit-equality of O and LSE, then
HIP-event timing of both forms alternating (20 launches each, three rounds)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_flamingo_amd import ops
dev = torch.device("cuda")


from audio_flamingo_amd import _lib


def run(persist, qkv, B, S, Hq, Hkv, D, causal):
    # persist: False = grid form, True = resident blocks + queue, "paired" = block k runs items k and total - 1 - k (afk_attn_set_persist_paired)
    ops.ATTN_PERSIST = bool(persist)
    _lib.call("afk_attn_set_persist_paired", int(persist == "paired"))
    try:
        return ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
    finally:
        _lib.call("afk_attn_set_persist_paired", 0)


for name, B, S, Hq, Hkv, D, causal in [("decoder S=1024", 8, 1024, 28, 4, 128, True), ("decoder S=2048", 4, 2048, 28, 4, 128, True), ("long S=7808", 1, 7808, 28, 4, 128, True),
                                       ("encoder-like S=1536 D=64", 8, 1536, 20, 20, 64, False), ("small", 1, 128, 4, 2, 128, True)]:
    qkv = (torch.randn((B * S, (Hq + 2 * Hkv) * D), device=dev) * 0.5).to(torch.bfloat16)
    o0, l0 = run(False, qkv, B, S, Hq, Hkv, D, causal)
    eq = []
    for _ in range(3):
        o1, l1 = run(True, qkv, B, S, Hq, Hkv, D, causal)
        torch.cuda.synchronize()
        eq.append(bool(torch.equal(o0, o1)) and bool(torch.equal(l0[..., :S], l1[..., :S])))
    q = ops._attn_queue(dev)
    row = {"shape": name, "bit_equal": eq, "queue_after": q.tolist()}
    op, lp = run("paired", qkv, B, S, Hq, Hkv, D, causal)
    torch.cuda.synchronize()
    row["paired_bit_equal"] = bool(torch.equal(o0, op)) and bool(torch.equal(l0[..., :S], lp[..., :S]))
    ts = {False: [], True: [], "paired": []}
    for rnd in range(3):
        for persist in (False, True, "paired"):
            for _ in range(3):
                run(persist, qkv, B, S, Hq, Hkv, D, causal)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run(persist, qkv, B, S, Hq, Hkv, D, causal)
            e1.record()
            torch.cuda.synchronize()
            ts[persist].append(round(1000 * e0.elapsed_time(e1) / 20, 1))
    row["grid_us"], row["persistent_us"], row["paired_us"] = ts[False], ts[True], ts["paired"]
    print(json.dumps(row), flush=True)
