"""Where does the time of one decode-chain launch go?  (needs a `make PROBES=1` build; AFK_LIB_PATH=<that libafk.so>)   python probe_decode_chain.py qkv|o_proj
Lane 0 of every wave stamps the 100 MHz wall clock at: 0 entry, 1 row statistic known (PRO_RMS: the input row is back), 2 first chunk multiplied (its weights
are back), 3 all chunks multiplied and the lanes reduced, 4 past the block barrier.  The caches are flushed before every launch."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from audio_flamingo_amd import _lib, ops

assert _lib.has_probes(), "needs a -DAFK_PROBES build (AFK_LIB_PATH)"
which = sys.argv[1] if len(sys.argv) > 1 else "qkv"
dev = torch.device("cuda")
BF = torch.bfloat16
H, Hq, Hkv, D, I = 3584, 28, 4, 128, 18944
nq, nk = Hq * D, Hkv * D
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=0.02: (torch.randn(*s, device=dev, generator=g) * sc).to(BF)
x, x2, q, o = rnd(1, H, sc=1.0), torch.empty(1, H, device=dev, dtype=BF), torch.empty(1, nq, device=dev, dtype=BF), rnd(1, nq, sc=1.0)
nw, bias = rnd(H, sc=1.0), rnd(nq + 2 * nk, sc=0.1)
Wqkv, Wo, Wgu = rnd(nq + 2 * nk, H), rnd(H, nq), rnd(2 * I, H)
act = torch.empty(1, I, device=dev, dtype=BF)
Smax = 1024
spad = ops.pad64(Smax)
Kc, Vt = rnd(1, Smax, nk, sc=1.0), rnd(1, Hkv, D, spad, sc=1.0)
cos, sin = rnd(Smax, D, sc=1.0), rnd(Smax, D, sc=1.0)
pos = torch.tensor([800], device=dev, dtype=torch.int32)
flush = torch.empty(1 << 28, device=dev, dtype=torch.uint8)
stamps = torch.zeros(40000 * 8, device=dev, dtype=torch.int64)
lib = _lib.load()
lib.afk_probe_decode_chain_stamps.argtypes = [ctypes.c_void_p]
assert lib.afk_probe_decode_chain_stamps(stamps.data_ptr()) == 0
st = ops._stream()
rows = []
for it in range(10):
    flush.fill_(it)
    stamps.zero_()
    torch.cuda.synchronize()
    if which == "qkv":
        _lib.call("afk_decode_chain_qkv", x.data_ptr(), nw.data_ptr(), 1e-6, Wqkv.data_ptr(), H, H, bias.data_ptr(), cos.data_ptr(), sin.data_ptr(), pos.data_ptr(), q.data_ptr(), Kc.data_ptr(), Vt.data_ptr(), spad, pos.data_ptr(), Hq, Hkv, D, st)
    elif which == "o_proj":
        _lib.call("afk_decode_chain_linear_residual", o.data_ptr(), Wo.data_ptr(), nq, H, nq, x.data_ptr(), x2.data_ptr(), st)
    else:
        _lib.call("afk_decode_chain_gate_up", x.data_ptr(), nw.data_ptr(), 1e-6, Wgu.data_ptr(), H, I, H, act.data_ptr(), st)
    torch.cuda.synchronize()
    s = stamps.view(-1, 8).cpu().double() * 0.01
    s = s[s[:, 0] > 0]
    t0 = s[:, 0].min()
    pct = lambda v: [round(float(v.quantile(qq)), 2) for qq in (0.1, 0.5, 0.9, 1.0)]
    rows.append({"waves": int(s.shape[0]), "entry_after_first_us(p10,p50,p90,max)": pct(s[:, 0] - t0),
                 "phase_us(p10,p50,p90,max)": {n: pct(s[:, k + 1] - s[:, k]) for k, n in enumerate(["->stat", "->chunk0", "->reduced", "->barrier"])},
                 "waves_entered_in_first_2us": int(((s[:, 0] - t0) < 2.0).sum()),
                 "first_entry_to_last_barrier_us": round(float(s[:, 4].max() - t0), 2)})
lib.afk_probe_decode_chain_stamps(None)
print(json.dumps({"launch": which, "launches": rows[2:5]}))
