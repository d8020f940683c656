"""SwiGLU backward (afk_silu_mul_bwd) at the AF3 decoder shape: HIP-event timing, TB/s on the algorithmic bytes (dh + gate|up read, d(gate|up) written);
AFK_SILU_BWD_FORM=1 / 2 selects the kernel form (run once per form; a checksum of the output is printed so that the forms can be compared)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_flamingo_amd import ops
dev = torch.device("cuda")
rows, I = 8192, 18944
gu = (torch.randn((rows, 2 * I), device=dev)).to(torch.bfloat16)
dh = (torch.randn((rows, I), device=dev) * 0.1).to(torch.bfloat16)
out = ops.silu_mul_bwd(gu, dh)
torch.cuda.synchronize()
ts = []
for rnd in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.silu_mul_bwd(gu, dh)
    e1.record()
    torch.cuda.synchronize()
    ts.append(1000 * e0.elapsed_time(e1) / 20)
nbytes = 2 * rows * (I + 2 * I + 2 * I)
print(json.dumps({"form": os.environ.get("AFK_SILU_BWD_FORM", "default"), "us": [round(t, 1) for t in ts], "TB_per_s": round(nbytes / min(ts) / 1e6, 2),
                  "checksum": [float(out.float().sum()), float(out.float().abs().sum())]}))
