"""PROBE (wrong results by design; needs the -DAFK_PROBES build: make PROBES=1 OUT=../lib_probes, AFK_LIB_PATH=<that libafk.so>): how much of the forward / dQ
tile loop is waiting for the LDS-DMA prefetch?  AFK_ATTN_DBG bit 0 = the fast tiles issue NO LDS-DMA (they re-read stale tiles), bit 1 = the tile barrier does
not wait for the prefetch (vmcnt), bit 4 (16) = no block barrier in the fast tiles at all (each wave only waits for its own DMA pieces).  HIP-event time of the forward alone per setting; one process per setting (the flag is read once)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    from audio_flamingo_amd import ops
    dev = torch.device("cuda")
    out = {}
    for name, B, S, Hq, Hkv, D, causal in [("decoder S=1024", 8, 1024, 28, 4, 128, True), ("encoder S=1500 D=64", 8, 1500, 20, 20, 64, False), ("5-min S=7774", 1, 7774, 28, 4, 128, True)]:
        qkv = (torch.randn((B * S, (Hq + 2 * Hkv) * D), device=dev) * 0.5).to(torch.bfloat16)
        for _ in range(3):
            ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
        ts = []
        n = 20 if S < 4000 else 6
        for rnd in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                ops.attn_fwd(qkv, B, S, Hq, Hkv, D, scale=D ** -0.5, causal=causal)
            e1.record()
            torch.cuda.synchronize()
            ts.append(round(1000 * e0.elapsed_time(e1) / n, 1))
        out[name] = ts
    print(json.dumps({"AFK_ATTN_DBG": os.environ.get("AFK_ATTN_DBG", "0"), "fwd_us": out}))
else:
    for dbg in ("0", "32", "33", "49"):
        env = dict(os.environ, AFK_ATTN_DBG=dbg, AFK_LIB_PATH=os.path.join(ROOT, "audio-flamingo_amd", "lib_probes", "libafk.so"))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=600)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:], flush=True)
