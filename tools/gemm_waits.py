"""Where do the waves of the free-running GEMM (gemm256f8.hip, instrumented build = variant 12) spend their time?
s_memtime stamps around the per-K-tile wait: per wave {loop cycles, cycles in lgkmcnt(0)+vmcnt(12), cycles in the barrier}.
    python tools/gemm_waits.py [M N K]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_flamingo_amd import _lib, ops

shapes = [(8192, 37888, 3584), (8192, 3584, 18944), (8192, 8192, 8192), (12000, 5120, 1280)]
if len(sys.argv) == 4:
    shapes = [tuple(int(x) for x in sys.argv[1:4])]
dev = torch.device("cuda")
for M, N, K in shapes:
    a = (torch.rand((M, K), device=dev) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand((N, K), device=dev) * 2 - 1).to(torch.bfloat16)
    c = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    nblk = ((M + 255) // 256) * ((N + 255) // 256)
    dbg = torch.zeros(nblk * 8 * 4 + nblk * 6 * 2, device=dev, dtype=torch.float32)
    ops.gemm_set_variant(12)
    for _ in range(2):
        _lib.call("afk_gemm_nt_bf16", a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), c.stride(0), M, N, K, 0, 0, 0, 0,
                  dbg.data_ptr(), 1.0, 0, ops._stream())
    torch.cuda.synchronize()
    ops.gemm_set_variant(0)
    d = dbg[: nblk * 32].reshape(nblk * 8, 4).double()
    tl = dbg[nblk * 32:].view(torch.float64).reshape(nblk, 6).cpu()
    loop, vm, bar, T = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
    # block timelines (100 MHz ticks -> us); gap = idle time of a CU between the end of one workgroup and the entry of the next
    ent, pro, lp, end, hw, xcc = [tl[:, i] for i in range(6)]
    cu = (xcc.long() << 32) | (hw.long() & 0xFFFFFFF0)  # drop the wave-slot bits, keep simd?/cu/sh/se
    gaps = []
    for c in cu.unique():
        idx = (cu == c).nonzero().flatten()
        o = idx[ent[idx].argsort()]
        if len(o) > 1:
            gaps.append(((ent[o][1:] - end[o][:-1]) / 100.0))
    gaps = torch.cat(gaps) if gaps else torch.zeros(1)
    tim = {"prologue_us": float(((pro - ent) / 100).mean()), "loop_us": float(((lp - pro) / 100).mean()), "epilogue_us": float(((end - lp) / 100).mean()),
           "block_us": float(((end - ent) / 100).mean()), "gap_between_blocks_us_mean": float(gaps.mean()), "gap_p50": float(gaps.median()),
           "n_cu_ids": int(cu.unique().numel()), "kernel_span_us": float((end.max() - ent.min()) / 100)}
    ideal = T * 16 * 2 * 32  # 16 MFMAs per wave and tile, two waves per SIMD, 32 cycles each
    print(json.dumps({"shape": [M, N, K], "loop_cycles_mean": float(loop.mean()), "ideal_mfma_cycles": float(ideal.mean()),
                      "mfma_busy_in_loop": float((ideal / loop).mean()), "frac_in_vmcnt_wait": float((vm / loop).mean()),
                      "frac_in_barrier": float((bar / loop).mean()), "vm_wait_cycles_per_tile": float((vm / T).mean()),
                      "barrier_cycles_per_tile": float((bar / T).mean()), "worst_wave_vm_frac": float((vm / loop).max()), **tim}), flush=True)
