"""issue pacing of v_mfma_f32_32x32x16_bf16 for one wave per SIMD: independent accumulators (mode 2) vs four dependent MFMAs in a row (mode 3) -
afk_mfma_ceiling modes 2 / 3, cycles per MFMA from s_memtime.  Few blocks (no power cap), N(0,1) operands."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from audio_flamingo_amd import _lib
dev = torch.device("cuda")
ops_ = torch.randn(65536, device=dev).to(torch.bfloat16)
sink = torch.zeros(4, device=dev, dtype=torch.float32)
out = {}
for mode, name in ((2, "round_robin_4_accumulators"), (3, "4_dependent_in_a_row")):
    for nb in (8, 256):
        vals = []
        for _ in range(3):
            _lib.call("afk_mfma_ceiling", mode, nb, 2000, ops_.data_ptr(), sink.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            vals.append(round(float(sink[0]), 2))
        out[f"{name}_blocks{nb}"] = vals
print(json.dumps({"cycles_per_mfma": out}))
