"""GEMM microbenchmark on the AF3-7B shapes (random data, guide rule 25): the two 256x256 NT kernels (v2 = 8-wave ping-pong, v3 = 4-wave
128x128 per wave), HIP-event timed through afk_prof_*.   python tools/bench_gemm.py [zeros]   (zeros: zero-filled operands = DVFS probe)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_flamingo_amd import ops

SHAPES = [  # (name, M, N, K)
    ("dec gate_up fwd", 8192, 37888, 3584), ("dec down fwd", 8192, 3584, 18944), ("dec qkv fwd", 8192, 4608, 3584), ("dec o fwd", 8192, 3584, 3584),
    ("dec gate_up wgrad", 37888, 3584, 8192), ("dec down wgrad", 3584, 18944, 8192), ("dec gate_up dgrad", 8192, 3584, 37888),
    ("lm_head chunk", 2048, 152064, 3584), ("lm_head dgrad", 2048, 3584, 152064), ("lm_head wgrad", 152064, 3584, 2048),
    ("enc qkv fwd", 12000, 3840, 1280), ("enc fc1 fwd", 12000, 5120, 1280), ("enc fc2 fwd", 12000, 1280, 5120), ("enc out fwd", 12000, 1280, 1280),
    ("enc fc1 wgrad", 5120, 1280, 12032), ("enc qkv wgrad", 3840, 1280, 12032), ("enc out wgrad", 1280, 1280, 12032),
    ("square 4096", 4096, 4096, 4096), ("square 8192", 8192, 8192, 8192),
]
dev = torch.device("cuda")
ZEROS = len(sys.argv) > 1 and sys.argv[1] == "zeros"
if len(sys.argv) > 1 and sys.argv[1] == "nt":
    SHAPES = SHAPES[:12] + SHAPES[-2:]
if len(sys.argv) > 2 and sys.argv[2] == "few":
    SHAPES = [SHAPES[0], SHAPES[1], SHAPES[3], SHAPES[11], SHAPES[-1]]
res = []
if len(sys.argv) > 1 and sys.argv[1] in ("bw", "bwzeros"):
    SHAPES = []
for name, M, N, K in SHAPES:
    a = (torch.rand((M, K), device=dev) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand((N, K), device=dev) * 2 - 1).to(torch.bfloat16)
    if ZEROS:
        a.zero_(); b.zero_()
    c = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    row = {"name": name, "M": M, "N": N, "K": K}
    for v in [int(x) for x in os.environ.get("AFK_BENCH_VARIANTS", "2,3").split(",")]:
        ops.gemm_set_variant(v)
        for _ in range(2):
            ops.gemm_nt(a, b, out=c)
        torch.cuda.synchronize()
        ops.prof_reset(); ops.prof_enable(True)
        for _ in range(5):
            ops.gemm_nt(a, b, out=c)
        ops.prof_enable(False)
        ms, fl, n = ops.prof_collect()
        row[f"v{v}_tflops"] = round(fl / ms / 1e9, 1)
        row[f"v{v}_us"] = round(1e3 * ms / n, 1)
    ops.gemm_set_variant(0)
    res.append(row)
    print(json.dumps(row), flush=True)

if len(sys.argv) > 1 and sys.argv[1] not in ("bw", "bwzeros"):
    sys.exit(0)
ZEROS = ZEROS or (len(sys.argv) > 1 and sys.argv[1] == "bwzeros")
# ---- backward forms: NN (dgrad) and TN (wgrad) against the NT kernel fed with pre-transposed operands
print("--- NN / TN forms (v2 = 256 kernels); nt_* = same contraction on the NT kernel incl. nothing else", flush=True)
BW = [("dec gate_up dgrad NN", "NN", 8192, 3584, 37888), ("dec down dgrad NN", "NN", 8192, 18944, 3584), ("dec qkv dgrad NN", "NN", 8192, 3584, 4608),
      ("lm_head dgrad NN", "NN", 2048, 3584, 152064), ("enc fc1 dgrad NN", "NN", 12000, 1280, 5120), ("enc qkv dgrad NN", "NN", 12000, 1280, 3840),
      ("dec gate_up wgrad TN", "TN", 37888, 3584, 8192), ("dec down wgrad TN", "TN", 3584, 18944, 8192), ("dec qkv wgrad TN", "TN", 4608, 3584, 8192),
      ("lm_head wgrad TN", "TN", 152064, 3584, 2048), ("enc fc1 wgrad TN", "TN", 5120, 1280, 12000), ("enc qkv wgrad TN", "TN", 3840, 1280, 12000),
      ("enc out wgrad TN", "TN", 1280, 1280, 12000), ("proj l2 wgrad TN", "TN", 3584, 3584, 6000)]
for name, form, M, N, K in BW:
    if form == "NN":
        a = (torch.rand((M, K), device=dev) * 2 - 1).to(torch.bfloat16)
        b = (torch.rand((K, N), device=dev) * 2 - 1).to(torch.bfloat16)
        kw = dict(trans_b=True)
    else:
        a = (torch.rand((K, M), device=dev) * 2 - 1).to(torch.bfloat16)
        b = (torch.rand((K, N), device=dev) * 2 - 1).to(torch.bfloat16)
        kw = dict(trans_a=True, trans_b=True)
    if ZEROS:
        a.zero_(); b.zero_()
    c = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    for _ in range(2):
        ops.gemm(a, b, out=c, **kw)
    torch.cuda.synchronize()
    ops.prof_reset(); ops.prof_enable(True)
    for _ in range(5):
        ops.gemm(a, b, out=c, **kw)
    ops.prof_enable(False)
    ms, fl, n = ops.prof_collect()
    print(json.dumps({"name": name, "M": M, "N": N, "K": K, "tflops": round(fl / ms / 1e9, 1), "us": round(1e3 * ms / n, 1)}), flush=True)
