"""generate() on the full AF3-7B geometry (random-init): prefill time and per-token decode time against the HBM roofline
(decode reads every decoder weight once per token: 14.1 GB bf16 -> 1.8 ms at 8 TB/s).   python tools/bench_decode.py [B]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from audio_flamingo_amd.frontend import LogMelFrontend
from audio_flamingo_amd.modeling import AudioFlamingo3ForConditionalGeneration

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda")
model = AudioFlamingo3ForConditionalGeneration(bench.af3_7b_config(), device=dev, init_seed=0)
model.check_placeholders = False
waves, ids, _ = bench.synthetic_batch(B, 0, dev)
ids = ids[:, : 9 + 750 + 9]  # prompt only: 9 + 750 <sound> + 9
feats = LogMelFrontend(dev)(waves, out_dtype=torch.bfloat16)
res = {"batch": B, "prompt_tokens": ids.shape[1]}
for new in (1, 33):
    model.generate(ids, input_features=feats, max_new_tokens=new)  # warm
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.generate(ids, input_features=feats, max_new_tokens=new)
    torch.cuda.synchronize()
    res[f"t_{new}"] = time.perf_counter() - t0
res["prefill_plus_first_token_ms"] = 1e3 * res["t_1"]
res["decode_ms_per_token"] = 1e3 * (res["t_33"] - res["t_1"]) / 32
res["decode_tokens_per_s"] = B / (res["decode_ms_per_token"] * 1e-3)
res["hbm_roofline_ms_per_token"] = 14.1e9 / 8e12 * 1e3
print(json.dumps(res))
