"""per-wave cycle accounting of the dK/dV sweep (AFK_ATTN_DBG=4 [+8]: s_memtime stamps inside attn_bwd_dkdv_lds_kernel; the sweep writes
{loop cycles, body cycles, barrier cycles, tiles, loop ticks of the 100 MHz clock, phase-1 cycles, key block} per wave into the dQ
buffer and the dQ / GQA-reduce kernels are skipped - gradients are garbage in this mode)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("AFK_ATTN_DBG", "4")
import torch
from audio_flamingo_amd import ops
dev = torch.device("cuda")
shapes = {"decoder": (8, 1024, 28, 4, 128, True), "encoder": (8, 1500, 20, 20, 64, False)}
for name in sys.argv[1:] or ["decoder"]:
    B, S, Hq, Hkv, D, causal = shapes[name]
    qkv = (torch.randn((B * S, (Hq + 2 * Hkv) * D), device=dev) * 0.5).to(torch.bfloat16)
    do = (torch.randn((B * S, Hq * D), device=dev) * 0.5).to(torch.bfloat16)
    o, lse = ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
    for _ in range(3):
        dqkv = ops.attn_bwd(qkv, o, do, lse, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
    torch.cuda.synchronize()
    nblk = ((S + 127) // 128) * Hq * B
    w = dqkv.view(-1).view(torch.float32)[: nblk * 4 * 8].view(nblk, 4, 8).double().cpu()
    loop, body, bar, tiles, ticks, ph1 = (w[..., i] for i in range(6))
    T = tiles.sum()
    clk = (loop.sum() / ticks.sum() * 100.0).item()   # MHz of the shader clock during the loops
    print(json.dumps({"shape": name, "dbg": os.environ["AFK_ATTN_DBG"], "blocks": nblk, "tiles_per_block": round((T / nblk / 4).item(), 2),
                      "shader_clock_MHz": round(clk, 1),
                      "cycles_per_tile": round((loop.sum() / T).item(), 1), "body": round((body.sum() / T).item(), 1),
                      "phase1": round((ph1.sum() / T).item(), 1), "barrier_wait": round((bar.sum() / T).item(), 1),
                      "mfma_cycles_per_tile": 64 * 32 if D == 128 else 32 * 32}), flush=True)
    if Hq == Hkv:
        continue  # without the GQA scratch the sweep writes dK/dV into dqkv, over the timeline records
    # block timeline (100 MHz ticks -> us): prologue / loop / epilogue per block, idle gaps per CU
    tl = dqkv.view(-1).view(torch.float32)[nblk * 4 * 8: nblk * 4 * 8 + nblk * 12].view(torch.float64).view(nblk, 6).cpu()
    ent, ls, le, end = (tl[:, i] / 100.0 for i in range(4))
    cu = (tl[:, 5].long() << 8) | ((tl[:, 4].long() >> 8) & 0xff)
    t0, t1 = ent.min().item(), end.max().item()
    gaps, first, last, per_cu = [], [], [], []
    for c in cu.unique().tolist():
        m = cu == c
        e, x = ent[m], end[m]
        o = e.argsort()
        e, x = e[o], x[o]
        # blocks of one CU may overlap when two are resident (head_dim 64): idle = time covered by no block
        cover, cur_end = 0.0, t0
        for a, b in zip(e.tolist(), x.tolist()):
            if b > cur_end:
                cover += b - max(a, cur_end)
                cur_end = b
        per_cu.append(cover)
        first.append(e[0].item() - t0)
        last.append(t1 - x.max().item())
    n_cu = len(per_cu)
    print(json.dumps({"shape": name, "kernel_us": round(t1 - t0, 1), "cus": n_cu, "blocks_per_cu": round(nblk / n_cu, 2),
                      "prologue_us": round((ls - ent).mean().item(), 2), "loop_us": round((le - ls).mean().item(), 2),
                      "epilogue_us": round((end - le).mean().item(), 2),
                      "cu_covered_frac": round(sum(per_cu) / n_cu / (t1 - t0), 3),
                      "first_block_delay_us": round(sum(first) / n_cu, 2), "tail_idle_us_mean": round(sum(last) / n_cu, 2),
                      "tail_idle_us_max": round(max(last), 2)}), flush=True)
