"""Round-4 probe of VERDICT r03 item 3a (needs a `make PROBES=1` build; AFK_LIB_PATH=<that libafk.so>): the persistent 256x256 tile loop with the next
tile's X0 Y0 X1 LDS-DMA issued AHEAD of the epilogue stores (variant 14, tools/probes/gemm256q.hip) against the one-tile-per-workgroup product kernel
(variant 2) and the round-2 persistent loop that issued them behind the stores (variant 13).  Interleaved rounds, HIP-event timed through afk_prof_*;
the variants must agree bit for bit (same per-tile arithmetic) - which is also the check of the in-order vmcnt assumption variant 14 rests on."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from audio_flamingo_amd import ops, _lib

assert _lib.has_probes(), "needs a -DAFK_PROBES build (AFK_LIB_PATH)"
dev = torch.device("cuda")
SHAPES = [("gate|up forward", 8192, 37888, 3584), ("down-proj forward", 8192, 3584, 18944), ("encoder fc1 (K = 1280, ragged M)", 12000, 5120, 1280),
          ("encoder fc1, M padded to whole tiles", 12032, 5120, 1280), ("decoder qkv", 8192, 4608, 3584), ("square 8192", 8192, 8192, 8192)]
out = []
for name, M, N, K in SHAPES:
    a = (torch.rand((M, K), device=dev) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand((N, K), device=dev) * 2 - 1).to(torch.bfloat16)
    row = {"shape": name, "M": M, "N": N, "K": K, "tiles": ((M + 255) // 256) * ((N + 255) // 256)}
    ref = None
    for rnd in range(3):
        for v in (2, 13, 14):
            ops.gemm_set_variant(v)
            c = ops.gemm_nt(a, b)
            torch.cuda.synchronize()
            if v == 2 and ref is None:
                ref = c.clone()
            elif rnd == 0:
                row[f"v{v}_bit_identical"] = bool(torch.equal(c, ref))
            ops.prof_reset(); ops.prof_enable(True)
            for _ in range(6):
                ops.gemm_nt(a, b, out=c)
            ops.prof_enable(False)
            ms, fl, n = ops.prof_collect()
            row.setdefault(f"v{v}_us", []).append(round(1e3 * ms / n, 1))
            row.setdefault(f"v{v}_tflops", []).append(round(fl / ms / 1e9))
    ops.gemm_set_variant(0)
    # repeated launches of variant 14 must agree with each other too (a race between the early DMA and late fragment reads would show here)
    ops.gemm_set_variant(14)
    c0 = ops.gemm_nt(a, b)
    row["v14_deterministic"] = all(bool(torch.equal(ops.gemm_nt(a, b), c0)) for _ in range(4))
    ops.gemm_set_variant(0)
    out.append(row)
    print(json.dumps(row), flush=True)
