"""Where do the ~11 us of the single-sequence decode attention go?  (needs a `make PROBES=1` build; AFK_LIB_PATH=<that libafk.so>)
Thread 0 of every block stamps the 100 MHz wall clock at: 0 entry, 1 key range known, 2 scores + block max done (all K loads back), 3 softmax sum done,
4 P.V reduced, 5 partials stored + drained + barrier, 6 counter bumped (+ barrier), 7 merge written (last block of a head only)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from audio_flamingo_amd import _lib, ops

assert _lib.has_probes(), "needs a -DAFK_PROBES build (AFK_LIB_PATH)"
dev = torch.device("cuda")
BF = torch.bfloat16
Hq, Hkv, D = 28, 4, 128
nq, nk = Hq * D, Hkv * D
keys = int(sys.argv[1]) if len(sys.argv) > 1 else 800
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ns = int(os.environ.get("NS", "8"))
Smax = max(1024, keys + 64)
spad = ops.pad64(Smax)
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g).to(BF)
NSET = 6
Kc = [rnd(B, Smax, nk) for _ in range(NSET)]
Vt = [rnd(B, Hkv, D, spad) for _ in range(NSET)]
flush = torch.empty(1 << 28, device=dev, dtype=torch.uint8)   # 256 MiB: evicts L2 and the memory-side cache between launches
q, o = rnd(B, nq), torch.empty(B, nq, device=dev, dtype=BF)
kr = torch.tensor([[0, keys + 1]] * B, device=dev, dtype=torch.int32)
aws = torch.zeros(_lib.load().afk_attn_decode_workspace_floats(B, Hq, D, ns), device=dev, dtype=torch.float32)
stamps = torch.zeros(B * Hq * ns * 8, device=dev, dtype=torch.int64)
lib = _lib.load()
lib.afk_probe_attn_decode_stamps.argtypes = [ctypes.c_void_p]
assert lib.afk_probe_attn_decode_stamps(stamps.data_ptr()) == 0
st = ops._stream()
rows = []
for it in range(12):
    flush.fill_(it)
    stamps.zero_()
    torch.cuda.synchronize()
    i = it % NSET
    _lib.call("afk_attn_decode_fused", q.data_ptr(), nq, D, Kc[i].data_ptr(), Smax * nk, nk, D, Vt[i].data_ptr(), Hkv * D * spad, spad, o.data_ptr(), nq, D,
              kr.data_ptr(), B, Hq, Hkv, D, float(D ** -0.5), ns, aws.data_ptr(), st)
    torch.cuda.synchronize()
    s = stamps.view(B * Hq * ns, 8).cpu().double() * 0.01   # us
    t0 = s[:, 0].min()
    last = s[:, 7] > 0
    rel = s - t0
    rows.append({"first_block_entry_to_last_block_entry_us": float(rel[:, 0].max()),
                 "block_entry_quantiles_us": [round(float(rel[:, 0].quantile(qq)), 2) for qq in (0.25, 0.5, 0.75, 0.9)],
                 "block_life_median_max_us": [round(float((s[:, 6] - s[:, 0]).median()), 2), round(float((s[:, 6] - s[:, 0]).max()), 2)],
                 "median_block_phase_us": [round(float((s[:, k + 1] - s[:, k]).median()), 2) for k in range(6)],
                 "last_blocks_merge_us": round(float((s[last, 7] - s[last, 6]).median()), 2),
                 "first_entry_to_last_stamp_us": round(float((s.max() - t0)), 2)})
lib.afk_probe_attn_decode_stamps(None)
print(json.dumps({"B": B, "keys": keys, "nsplit": ns, "phases": "entry->range, ->scores, ->softmax, ->PV, ->stored, ->counted", "launches": rows[2:]}, indent=0))
