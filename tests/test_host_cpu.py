"""CPU: host-side logic - the C-ABI library loads and exports every declared symbol, the parameter arena / oracle-compatible
state_dict layout, loud failure without a GPU, and the data-parallel bucket exchange over gloo (world_size 2)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = dict(
    audio_config=dict(num_mel_bins=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256, hidden_size=128,
                      max_source_positions=1500),
    text_config=dict(vocab_size=1024, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                     num_key_value_heads=2, max_position_embeddings=4096),
    audio_token_id=1023,
)


def _cfg():
    from transformers import AudioFlamingo3Config

    return AudioFlamingo3Config(**TINY)


def test_decode_batch_routing_host_logic():
    """which batches take the one-launch-per-Linear decode step (host logic of modeling._chain_batch_cap / _prologue_shapes_ok, round 6): the tiny test geometry
    (hidden 256, 4:2 x 64 heads) and the AF3-7B geometry take the norm-in-prologue launches up to 8 sequences and groups of eight up to 32; AFK_DECODE_CHAIN_BATCH_MAX
    caps it; a geometry the 64-element blocks do not divide, or another norm mode, stays at eight sequences on rounds 4-5's launches"""
    import types

    import torch
    from audio_flamingo_amd.modeling import AudioFlamingo3ForConditionalGeneration as Mine

    m = Mine(_cfg(), device="cpu")
    head = m.arena["lm_head.weight"].data
    assert m._prologue_shapes_ok(head) and m._chain_batch_cap(head) == 32
    big = types.SimpleNamespace(H=3584, Hq=28, Hkv=4, D=128, I=18944, decode_chain_batch=8, decode_chain_batch_max=32, decode_norm_mode="prologue")
    big._prologue_shapes_ok = types.MethodType(Mine._prologue_shapes_ok, big)
    big._chain_batch_cap = types.MethodType(Mine._chain_batch_cap, big)
    wide = torch.empty((152064, 1), dtype=torch.bfloat16)
    assert big._prologue_shapes_ok(wide) and big._chain_batch_cap(wide) == 32
    big.decode_chain_batch_max = 16
    assert big._chain_batch_cap(wide) == 16
    big.decode_chain_batch_max, big.decode_norm_mode = 32, "launch"
    assert big._chain_batch_cap(wide) == 8
    big.decode_norm_mode, big.I = "prologue", 18944 + 8    # an intermediate size the 64-element blocks do not divide
    assert not big._prologue_shapes_ok(wide) and big._chain_batch_cap(wide) == 8


def test_library_exports_every_declared_symbol():
    from audio_flamingo_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    protos = _lib.prototypes()
    assert len(protos) >= 46
    lib = _lib.load()
    for name in protos:
        assert hasattr(lib, name), name
    assert lib.afk_version() >= 1
    # argument validation happens before any GPU work: exercise the error path without a device
    with pytest.raises(_lib.AfkError, match="null"):
        _lib.call("afk_gemm_nt_bf16", 0, 0, 0, 0, 0, 0, 1, 1, 64, 0, 0, 0, 0, 0, 1.0, 0, 0)


def test_round6_entry_points_validate_their_arguments_without_a_device():
    """the decode / backward entry points of round 6 refuse what they cannot run BEFORE any GPU work (the error text names the limit): more than 32 sequences,
    a reduction length the 64-element blocks do not divide, a partial-sum count that is not a multiple of four, null pointers"""
    from audio_flamingo_amd import _lib
    buf = torch.zeros(1 << 16, dtype=torch.float32)   # host memory: never touched - validation fails first
    p = buf.data_ptr()
    H, I = 3584, 18944
    with pytest.raises(_lib.AfkError, match="1 <= M <= 32"):
        _lib.call("afk_decode_chain_gate_up_norm_batched", p, H, 33, p, 1e-6, p, H, I, H, p, I, None, 0, 0)
    with pytest.raises(_lib.AfkError, match="K %% 64|K % 64"):
        _lib.call("afk_decode_chain_gate_up_norm_batched", p, H, 8, p, 1e-6, p, H, I, H + 8, p, I, None, 0, 0)
    with pytest.raises(_lib.AfkError, match="multiple of 4"):
        _lib.call("afk_decode_chain_gate_up_norm_batched", p, H, 8, p, 1e-6, p, H, I, H, p, I, p, 222, 0)
    with pytest.raises(_lib.AfkError, match="more than 8 sequences need"):
        _lib.call("afk_decode_chain_linear_residual_batched", p, H + 8, 12, p, H + 8, H, H + 8, p, H, p, H, 0)
    with pytest.raises(_lib.AfkError, match="1 <= M <= 32"):
        _lib.call("afk_decode_chain_linear_residual_ss_batched", p, H, 40, p, H, H, H, p, H, p, H, p, 0)
    with pytest.raises(_lib.AfkError, match="null"):
        _lib.call("afk_layernorm_bwd_colsum", p, p, p, p, p, p, p, p, p, 0, None, 0, p, 8, 64, 0)
    with pytest.raises(_lib.AfkError, match="C %% 8|C % 8"):
        _lib.call("afk_gelu_bwd_colsum", p, p, p, 8, 60, p, 0, p, 0)
    with pytest.raises(_lib.AfkError, match="mode"):
        _lib.call("afk_attn_decode_set_group", 7)
    assert _lib.load().afk_gelu_bwd_colsum_parts(12000) >= 256 and _lib.load().afk_gelu_bwd_colsum_parts(5) == 5


def test_state_dict_is_oracle_compatible_and_fused_views_alias():
    from audio_flamingo_amd.modeling import AudioFlamingo3ForConditionalGeneration as Mine

    sd = torch.load(os.path.join(ROOT, "tests", "golden", "tiny64_state_bf16.pt"))
    m = Mine(_cfg(), device="cpu")
    assert set(m.state_dict().keys()) == set(sd.keys())
    r = m.load_state_dict(sd)
    assert not r.missing_keys and not r.unexpected_keys
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k
    a = m.arena
    p = "model.language_model.layers.1."
    fused = a[p + "self_attn.qkv.weight"].data
    assert torch.equal(fused[:256], sd[p + "self_attn.q_proj.weight"]) and torch.equal(fused[256:384], sd[p + "self_attn.k_proj.weight"])
    gu = a[p + "mlp.gate_up.weight"].data
    assert torch.equal(gu[512:], sd[p + "mlp.up_proj.weight"])
    q = dict(m.named_parameters())[p + "self_attn.q_proj.weight"]
    assert q.data_ptr() == fused.data_ptr() and q.grad.data_ptr() == a[p + "self_attn.qkv.weight"].grad.data_ptr()
    # encoder k_proj has no bias: the middle third of the fused bias stays zero and is not a parameter
    eb = a["model.audio_tower.layers.0.self_attn.qkv.bias"].data
    assert (eb[128:256] == 0).all() and "model.audio_tower.layers.0.self_attn.k_proj.bias" not in m.state_dict()
    assert not dict(m.named_parameters())["model.audio_tower.embed_positions.weight"].requires_grad
    # buckets tile the arena in forward order, 16-byte aligned
    prev = 0
    for i in range(len(a.bucket_names)):
        s, e = a.bucket_range(i)
        assert s == prev and e > s and s % 64 == 0
        prev = e
    assert prev == a.total


def test_no_cpu_fallback():
    from audio_flamingo_amd._lib import AfkError
    from audio_flamingo_amd.modeling import AudioFlamingo3ForConditionalGeneration as Mine

    m = Mine(_cfg(), device="cpu")
    with pytest.raises(AfkError, match="HIP device"):
        m(input_ids=torch.zeros((1, 8), dtype=torch.long))


def test_bucket_ready_callbacks_fire_in_backward_order():
    from audio_flamingo_amd.modeling import AudioFlamingo3ForConditionalGeneration as Mine

    m = Mine(_cfg(), device="cpu")
    a = m.arena
    fired = []
    a.on_bucket_ready = fired.append
    a.zero_grad()
    for blk in reversed(a.order):
        a.grad_written(blk)
    assert fired == list(reversed(range(len(a.bucket_names))))
    assert all(not b.fresh for b in a.order)
    a.zero_grad()
    assert all(b.fresh for b in a.order)


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from transformers import AudioFlamingo3Config
from audio_flamingo_amd.modeling import AudioFlamingo3ForConditionalGeneration as Mine
from audio_flamingo_amd.dp import DataParallelEngine
import tests.test_host_cpu as T
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
m = Mine(AudioFlamingo3Config(**T.TINY), device="cpu", init_seed=rank)   # different init per rank on purpose
a = m.arena
eng = DataParallelEngine(a)
eng.broadcast_parameters(0)
ref = Mine(AudioFlamingo3Config(**T.TINY), device="cpu", init_seed=0)
assert torch.equal(a.params, ref.arena.params), "broadcast failed"
# each rank writes rank-dependent gradients block by block in backward order; buckets are exchanged as they complete
a.zero_grad(); eng.begin_backward()
g = torch.Generator().manual_seed(100 + rank)
local = (torch.randn(a.total, generator=g) * 0.1).to(torch.bfloat16)
for blk in reversed(a.order):
    blk.grad.copy_(local[blk.offset: blk.offset + blk.numel].view(blk.shape))
    a.grad_written(blk)
eng.finish()
expect = torch.zeros(a.total)
for r in range(world):
    gr = torch.Generator().manual_seed(100 + r)
    expect += (torch.randn(a.total, generator=gr) * 0.1).to(torch.bfloat16).float()
for blk in a.order:
    got = blk.grad.float().reshape(-1)
    exp = expect[blk.offset: blk.offset + blk.numel]
    assert (got - exp).abs().max() <= 0.02, (blk.key, float((got - exp).abs().max()))
# no_sync: nothing is exchanged
a.zero_grad(); eng.begin_backward()
with eng.no_sync():
    for blk in reversed(a.order):
        blk.grad.fill_(float(rank + 1)); a.grad_written(blk)
    eng.finish()
assert float(a.order[0].grad.float().mean()) == float(rank + 1)
assert abs(eng.grad_scale - 1.0 / world) < 1e-12
# uneven step (ADVICE r01): rank 1's batch has no audio -> its backward never touches the audio tower / projector buckets.  Every rank
# must still issue the SAME collectives in the SAME order, stale slices must not leak in, and the "touched by any rank" gate must be 1.
audio = lambda b: b.key.startswith("model.audio_tower") or b.key.startswith("model.multi_modal_projector")
for blk in a.order:
    blk.grad.fill_(7.0)                      # stale values from "the previous step"
a.zero_grad(); eng.begin_backward()
for blk in reversed(a.order):
    if rank == 1 and audio(blk):
        continue
    blk.grad.fill_(1.0); a.grad_written(blk)
eng.finish()
orders = [None] * world
dist.all_gather_object(orders, list(eng.issued))
assert all(o == orders[0] for o in orders), orders
assert orders[0] == list(reversed(range(len(a.bucket_names)))), orders[0]
for blk in a.order:
    want = 1.0 if audio(blk) else float(world)
    assert float(blk.grad.float().min()) == want == float(blk.grad.float().max()), (blk.key, float(blk.grad.float().mean()))
assert eng.bucket_gate.tolist() == [1] * len(a.bucket_names)
# nobody touches the audio side (text-only step everywhere): gate 0 there -> the optimizer leaves those buckets alone
a.zero_grad(); eng.begin_backward()
for blk in reversed(a.order):
    if not audio(blk):
        blk.grad.fill_(1.0); a.grad_written(blk)
eng.finish()
gate = eng.bucket_gate.tolist()
for i, name in enumerate(a.bucket_names):
    assert gate[i] == (0 if (name.startswith("enc") or name == "stem") else 1), (name, gate[i])
dist.destroy_process_group()
print("OK", rank)
'''


def test_dp_bucket_allreduce_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(script), ROOT]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and r.stdout.count("OK") == 2, r.stdout[-2000:] + r.stderr[-3000:]


def test_hf_attention_plugin_registers_and_refuses_cpu():
    """the reference's plugin registry accepts the HIP attention; on CPU tensors the plugin refuses instead of falling back"""
    import pytest
    import torch
    from transformers import AttentionInterface

    from audio_flamingo_amd import hf_plugin

    name = hf_plugin.register()
    assert name in AttentionInterface()._global_mapping or name in AttentionInterface._global_mapping
    q = torch.zeros(1, 2, 4, 64, dtype=torch.bfloat16)
    with pytest.raises(TypeError):
        hf_plugin.afk_attention(torch.nn.Identity(), q, q, q, None)
    m = torch.ones(1, 1, 4, 4, dtype=torch.bool)
    m[0, 0, 2, 1] = False  # row 2 sees keys {0, 2, 3}: a hole
    with pytest.raises(NotImplementedError):
        hf_plugin._intervals(m)
    kr = hf_plugin._intervals(torch.ones(4, 4, dtype=torch.bool).tril()[None, None])
    assert kr.tolist() == [[[0, 1], [0, 2], [0, 3], [0, 4]]]


def test_music_flamingo_time_tables_match_oracle():
    """host-side index / angle arithmetic of the Music Flamingo delta (which window of its sample every encoder window is, cos / sin of the
    rotary time angles) against the oracle restatement, on CPU - the rotation kernel itself is covered by the GPU tests"""
    import torch
    from transformers import MusicFlamingoConfig

    from audio_flamingo_amd.musicflamingo import MusicFlamingoForConditionalGeneration as Mine
    from oracle import af3_oracle as O

    T = {k: (dict(v, model_type="audioflamingo3_encoder") if k == "audio_config" else v) for k, v in TINY.items()}
    m = Mine(MusicFlamingoConfig(**T), device="cpu")
    S = 9 + 1000 + 9 + 12
    ids = torch.randint(0, 1000, (2, S))
    ids[0, 9:1009] = 1023          # sample 0: two windows (750 + 250 tokens)
    ids[1, 9:134] = 1023           # sample 1: one window (125 tokens)
    post = torch.tensor([750, 250, 125])
    ts = m._audio_timestamps(ids, post, 750)
    ref_ts = O.music_audio_timestamps(ids, post, 750, 1023)
    assert torch.equal(ts, ref_ts)
    assert float(ts[1, 0]) == 30.0 and float(ts[2, 0]) == 0.0   # second window of sample 0 starts at 30 s; sample 1 restarts at 0
    # the window-index derivation (running placeholder count + "does the run continue" test) against the reference's run extraction on
    # layouts that stress it: a run ending at the last position of a row followed by a run starting at position 0 of the next row (two
    # samples, NOT one run), three windows in one sample, a sample without audio in between, ragged last windows
    S2 = 2000
    ids2 = torch.randint(0, 1000, (4, S2))
    ids2[0, S2 - 875:] = 1023            # sample 0: 750 + 125, run touches the END of its row
    ids2[1, :1750] = 1023                # sample 1: 750 + 750 + 250, run starts at position 0 of the next row
    ids2[3, 5:755] = 1023                # sample 2 (row 3; row 2 is text only): one full window
    post2 = torch.tensor([750, 125, 750, 750, 250, 750])
    ts2 = m._audio_timestamps(ids2, post2, 750)
    assert torch.equal(ts2, O.music_audio_timestamps(ids2, post2, 750, 1023))
    assert [float(x) for x in ts2[:, 0]] == [0.0, 30.0, 0.0, 30.0, 60.0, 0.0]
    cos, sin = m._tables(ts, 750)
    rc, rs = O.rotary_time_tables(ref_ts, 750, 128)
    assert cos.shape == rc.shape == (3, 750, 52)
    assert float((cos - rc).abs().max()) < 1e-5 and float((sin - rs).abs().max()) < 1e-5


def test_splitk_plans():
    from audio_flamingo_amd import ops

    assert ops.splitk_plan(8192, 3584, 3584) == 1          # enough 256x256 tiles: one pass
    assert ops.splitk_plan(1280, 1280, 12032) > 1          # encoder wgrad on the NT kernel: 100 tiles of 128x128
    assert ops.splitk_plan(1, 3584, 18944) > 1             # decode: down-projection
    assert ops.splitk_plan(1, 3584, 18944) <= (18944 + 511) // 512
    assert ops.splitk_plan(8, 3584, 256) == 1              # short reduction: nothing to share
    assert ops.splitk_plan_256(1280, 1280, 12000) > 1 and ops.splitk_plan_256(5120, 18944, 8192) == 1


def test_from_pretrained_and_save_pretrained_round_trip_with_the_reference(tmp_path):
    """checkpoint surface: a directory written by the REFERENCE's save_pretrained loads into the arena model (same names, same values),
    and a directory written by ours loads back into the reference class - single-file and sharded"""
    import torch
    from transformers import AudioFlamingo3Config, AudioFlamingo3ForConditionalGeneration

    from audio_flamingo_amd.modeling import AudioFlamingo3ForConditionalGeneration as Mine

    torch.manual_seed(0)
    ref = AudioFlamingo3ForConditionalGeneration(AudioFlamingo3Config(**TINY)).to(torch.bfloat16)
    ref.save_pretrained(tmp_path / "ref")
    m = Mine.from_pretrained(str(tmp_path / "ref"), device="cpu")
    sd_ref, sd = ref.state_dict(), m.state_dict()
    assert set(sd) == set(sd_ref)
    assert all(torch.equal(sd[k], sd_ref[k]) for k in sd_ref)
    m.save_pretrained(str(tmp_path / "ours"), max_shard_size=1 << 20)   # forces several shards
    back = AudioFlamingo3ForConditionalGeneration.from_pretrained(str(tmp_path / "ours"), dtype=torch.bfloat16)
    sd_back = back.state_dict()
    assert all(torch.equal(sd_back[k], sd_ref[k]) for k in sd_ref)


def test_library_is_a_build_of_this_tree():
    """libafk.so carries the sha256 prefix of the sources it was compiled from; the binding refuses any other tree (a GPU test run can
    therefore only exercise a build of the sources next to it)"""
    from audio_flamingo_amd import _lib

    lib = _lib.load()
    assert lib.afk_build_id().decode() == _lib.source_hash()


def test_peel_plan_for_nearly_empty_last_rounds():
    """the opt-in tail peel of the TN weight-gradient GEMMs: only shapes whose 256x256 tiles end in a nearly empty round get a plan, the
    strip is cut on a tile boundary and runs as one round of split-K workgroups"""
    from audio_flamingo_amd import ops

    axis, cut, sp = ops.peel_plan_256(37888, 3584, 8192)        # gate|up wgrad: 2 072 tiles = 8.09 rounds
    assert (axis, cut) == (0, 146 * 256) and 2 <= sp <= 16 and (37888 - cut) // 256 * 14 * sp <= 256
    axis, cut, sp = ops.peel_plan_256(3584, 18944, 8192)        # down wgrad: 1 036 tiles = 4.05 rounds
    assert (axis, cut) == (1, 73 * 256) and 14 * sp <= 256
    for shape in [(4608, 3584, 8192), (3584, 3584, 8192), (8192, 3584, 3584), (12000, 5120, 1280), (1280, 1280, 12000)]:
        assert ops.peel_plan_256(*shape) is None, shape


def test_trainer_unwraps_accelerate_optimizer_wrappers():
    """inside the HF Trainer loop `self.optimizer` is accelerate's AcceleratedOptimizer around AfkAdamW; the DP gates, the master
    sync and the clip request must reach the AfkAdamW underneath, through any depth of `.optimizer` wrappers"""
    from audio_flamingo_amd.trainer import AfkAdamW, unwrap_optimizer

    class Wrap:  # the shape of accelerate.optimizer.AcceleratedOptimizer that matters here
        def __init__(self, o):
            self.optimizer = o

    inner = AfkAdamW.__new__(AfkAdamW)   # no arena needed for the identity check
    assert unwrap_optimizer(inner) is inner
    assert unwrap_optimizer(Wrap(inner)) is inner
    assert unwrap_optimizer(Wrap(Wrap(inner))) is inner
    assert unwrap_optimizer(torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.1)) is None
    assert unwrap_optimizer(None) is None
    from accelerate.optimizer import AcceleratedOptimizer

    assert hasattr(AcceleratedOptimizer, "__init__") and "optimizer" in AcceleratedOptimizer.__init__.__code__.co_varnames


def test_bench_multi_rank_control_flow_gloo():
    """bench.py's N > 1 control flow, launched exactly as the driver launches it (python -m torch.distributed.run
    --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 ... bench.py --gpus 2 ...), over gloo on the host with a stub in place of the
    step (bench.py --dry-run-cpu): process group -> replica + parameter broadcast -> per-bucket all-reduces issued from the arena's ready
    callbacks in backward order -> replica checksum all-gather -> exactly ONE JSON line, from rank 0, as the LAST line of the output"""
    import json
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run-cpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    jl = [ln for ln in lines if ln.startswith("{")]
    assert len(jl) == 1 and lines[-1] == jl[0], lines[-5:]
    d = json.loads(jl[0])
    assert d["dry_run"] is True and d["value"] is None and d["n_gpus"] == 2 and d["replicas_identical_after_steps"] is True
    assert d["dp"]["preflight"]["torch"]["ok"] is True and d["dp"]["preflight"]["torch"]["failed"] == []   # known-answer all-reduce / reduce-scatter / all-gather / flag MAX / replica checksum
    assert d["buckets"] == 8 and d["collectives_per_step"] == 9    # stem, enc0, enc1, enc_out, embed, dec0, dec1, head + the touched-flag MAX


def test_bench_self_spawns_ranks_without_a_launcher():
    """`python bench.py --gpus 2 ...` with WORLD_SIZE unset (the command form the driver uses at N = 1) must not die on the
    world-size assert: it re-executes itself under torch.distributed.run with one rank per GPU and still ends on ONE parseable JSON line"""
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run-cpu"],
                       capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    d = json.loads(lines[-1])
    assert len(lines[-1]) < 4096 and d["dry_run"] is True and d["n_gpus"] == 2 and d["launched_by"] == "self_spawn" and d["replicas_identical_after_steps"] is True


def test_bench_line_stays_parseable_and_small():
    """BENCH_r05.json came back `parsed: null` because the line had grown to 30 KB.  The line is now a <= 4 KB summary built by
    bench.compact_line from the full record (which goes to gpurun_out/bench_detail.json); checked here on the largest record the repo holds - round 5's
    full default run with every leg, DP prose and the per-tensor parity table in it - and on a record whose optional parts are all oversized"""
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    with open(os.path.join(root, "profiles", "r05_bench_n1.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    d = json.loads(line)
    assert len(line) < bench.LINE_LIMIT <= 4096
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "roofline_hbm", "cpu_baseline", "parity_fulldepth"):
        assert k in d, k
    assert d["value"] == round(full["value"], 3) and d["ms_per_step"] == round(full["ms_per_step"], 3)
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and d["roofline"]["bound"] == "mfma"
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert d["parity_fulldepth"]["green"] is True and len(json.dumps(d["parity_fulldepth"])) <= 600
    assert "model" not in d["config"] and d["config"]["workload"]
    # pathological growth of the optional parts must shed them, never break the line
    fat = dict(full)
    fat["parity_fulldepth"] = dict(full["parity_fulldepth"], peaked={"note": "x" * 5000})
    line2 = bench.compact_line(fat, "gpurun_out/bench_detail.json")
    assert len(line2) < 4096 and json.loads(line2)["parity_fulldepth"]["green"] is True


def test_af3_output_is_lazy_about_logits():
    """forward(labels=...) hands back the reference's output surface with logits built on FIRST ACCESS: reading the loss (attribute, key or
    index 0 - what Trainer.compute_loss does) must not build them; outputs[1:] (what Trainer.prediction_step reads) must"""
    from audio_flamingo_amd.modeling import AF3Output

    calls = []
    out = AF3Output(loss=torch.tensor(1.5), logits_fn=lambda: (calls.append(1), torch.zeros(2, 3, 4))[1], audio_hidden_states=torch.ones(5))
    assert out[0] is out.loss and out["loss"] is out.loss and out.keys() == ["loss", "logits", "audio_hidden_states"] and len(out) == 3
    assert "logits" in out and "hidden_states" not in out and out.get("attentions", 7) == 7
    assert not calls and not out.logits_materialized
    rest = out[1:]
    assert calls == [1] and rest[0].shape == (2, 3, 4) and rest[1] is out.audio_hidden_states
    assert out.logits is rest[0] and out.to_tuple()[1] is rest[0] and calls == [1]        # built once
    with pytest.raises(KeyError):
        out["nope"]
    eager = AF3Output(logits=torch.zeros(1))
    assert eager.logits_materialized and eager.keys() == ["logits"] and eager[0] is eager.logits


def test_dp_flags_and_adamw_launch_plans():
    """host logic of round 3: (i) the per-bucket touched flags are written by fill launches (capturable into a HIP graph), one per run of
    ones; (ii) FusedAdamW's launch plans - flat runs merged per decay class, one fused launch per 2-D GEMM weight when the transposed-shadow
    kernel is enabled, none of those when the arena keeps lazy W^T shadows - and the per-block lazy shadow used by the lm_head"""
    from audio_flamingo_amd.arena import FusedAdamW
    from audio_flamingo_amd.dp import _flags_on_device
    from audio_flamingo_amd.modeling import AudioFlamingo3ForConditionalGeneration as Mine

    assert _flags_on_device([1, 1, 1], "cpu").tolist() == [1, 1, 1]
    assert _flags_on_device([0, 1, 1, 0, 1, 0], "cpu").tolist() == [0, 1, 1, 0, 1, 0]
    assert _flags_on_device([0, 0], "cpu").tolist() == [0, 0] and _flags_on_device([0, 0], "cpu").dtype == torch.int32
    m = Mine(_cfg(), device="cpu")
    opt = FusedAdamW(m.arena, lr=1e-3, weight_decay=0.01)
    assert not opt.fuse_shadow and all(op[0] == "flat" for op in opt.segments)
    covered = sum(op[2] - op[1] for op in opt.segments)
    assert covered == m.arena.total                                   # the flat runs tile the whole arena
    assert all(a[3] != b[3] for a, b in zip(opt.segments, opt.segments[1:]) if a[2] == b[1])   # adjacent runs differ in their decay class (else merged)
    opt.fuse_shadow = True                                            # plans are rebuilt when the policy changes
    kinds = [op[0] for op in opt.segments]
    t_keys = {op[1].key for op in opt.segments if op[0] == "T"}
    assert "T" in kinds and "flat" in kinds
    assert all(m.arena[k].shadow_kind == "T" and m.arena[k].shape[0] % 64 == 0 and m.arena[k].shape[1] % 64 == 0 for k in t_keys)
    assert "lm_head.weight" in t_keys and "model.language_model.embed_tokens.weight" not in t_keys      # embed_tokens has no dgrad shadow
    flat_cov = sum(op[2] - op[1] for op in opt.segments if op[0] == "flat")
    t_cov = sum((op[1].numel + 63) // 64 * 64 for op in opt.segments if op[0] == "T")
    assert flat_cov + t_cov == m.arena.total
    m.arena.lazy_T_shadows = True                                     # "direct" backward form: no eager shadows -> nothing to fuse
    assert all(op[0] == "flat" for op in opt.segments)
    m.arena.lazy_T_shadows = False
    blk = m.arena["lm_head.weight"]
    assert blk.shadow_lazy is False


def test_checkpoint_planner_recomputes_only_what_the_budget_requires():
    """memory-budgeted recompute (oracle switch: GradientCheckpointingLayer, modeling_layers.py:79-114): host arithmetic
    of the plan on the AF3-7B geometry - a 5-minute clip fits 228 GiB outright, a 10-minute clip recomputes part of the decoder, a small
    budget falls back to the reference's every-layer recompute"""
    import bench

    from audio_flamingo_amd.modeling import AudioFlamingo3ForConditionalGeneration as Mine

    m = Mine.__new__(Mine)   # geometry only: no arena, no device
    cfg = bench.af3_7b_config()
    ac, tc = cfg.audio_config, cfg.text_config
    m.config, m.enc_layers, m.dec_layers, m.max_pos, m.enc_heads = cfg, ac.num_hidden_layers, tc.num_hidden_layers, ac.max_source_positions, ac.num_attention_heads
    m.Hq, m.Hkv, m.D = tc.num_attention_heads, tc.num_key_value_heads, tc.hidden_size // tc.num_attention_heads
    m.ckpt_policy, m.ckpt_budget_bytes = "budget", None
    total = 288 * 10 ** 9
    resident = int(139.5 * 2 ** 30)          # parameters + gradients + AdamW state + W^T shadows of the 7B model
    enc5, dec5 = m.activation_bytes_per_layer(10, 7774)
    assert 0.055e9 * 10 < enc5 < 0.07e9 * 10 and 1.2e9 < dec5 < 1.3e9, (enc5, dec5)      # 61 MB per window-layer, 1.23 GB per decoder layer
    p5 = m.plan_checkpointing(10, 7774, resident, total)
    assert (p5["enc"], p5["dec"]) == (0, 0), p5
    p10 = m.plan_checkpointing(20, 15274, resident, total)
    assert p10["enc"] == 0 and 0 < p10["dec"] < 28, p10
    m.ckpt_budget_bytes = 150 * 2 ** 30
    tight = m.plan_checkpointing(20, 15274, resident, total)
    assert tight["dec"] == 28 and tight["enc"] == 32, tight
    m.ckpt_policy = "full"
    assert m.plan_checkpointing(10, 7774, resident, total)["dec"] == 28


NATIVE_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from audio_flamingo_amd import _lib, dp
# dry run of the C-ABI communicator's host protocol (no GPU here): _lib.call is replaced by a recorder that also plays rank 0's unique id back
calls = []
def fake_call(name, *args):
    calls.append((name, args))
    if name == "afk_comm_unique_id":
        import ctypes
        ctypes.memmove(args[0], bytes([17 + (i * 7) % 200 for i in range(128)]), 128)
    if name == "afk_comm_init":
        import ctypes
        ctypes.c_void_p.from_address(args[3]).value = 0xAF00 + args[0]
    return 0
_lib.call = fake_call
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
comm = dp.NativeComm.from_process_group()
ids = [None] * world
init = [c for c in calls if c[0] == "afk_comm_init"]
assert len(init) == 1 and init[0][1][0] == rank and init[0][1][1] == world, init
dist.all_gather_object(ids, init[0][1][2])
assert all(i == ids[0] for i in ids) and len(ids[0]) == 128, "every rank must initialise with rank 0's unique id"
assert sum(1 for c in calls if c[0] == "afk_comm_unique_id") == (1 if rank == 0 else 0)
assert comm.handle.value == 0xAF00 + rank
# the collectives go out with the communicator handle, the slice pointer, its element count and the dtype code of include/afk.h
import audio_flamingo_amd.ops as ops
ops._stream = lambda: 0
class T:   # stand-in for a device tensor (pointer + size is all the C ABI sees)
    is_cuda = True
    dtype = torch.bfloat16
    def __init__(self, n): self.n = n
    def is_contiguous(self): return True
    def data_ptr(self): return 0x1000
    def numel(self): return self.n
calls.clear()
comm.allreduce_(T(1000), form="rs_ag"); comm.allreduce_(T(1000), form="allreduce"); comm.broadcast_(T(77), root=0)
names = [c[0] for c in calls]
assert names == ["afk_reduce_scatter_allgather_bucket", "afk_allreduce_bucket", "afk_comm_broadcast"], names
assert calls[0][1][1:4] == (0x1000, 1000, 0) and calls[2][1][1:5] == (0x1000, 77, 0, 0), calls
comm.close()
assert calls[-1][0] == "afk_comm_destroy" and comm.handle is None
dist.destroy_process_group()
print("OK", rank)
'''


def test_native_comm_bootstrap_protocol_gloo_world2(tmp_path):
    """pre-flight of the first multi-rank run: the host protocol of the C-ABI RCCL communicator - rank 0 creates the unique id,
    every rank receives it over the torch.distributed group and calls afk_comm_init(rank, world, id); the bucket collectives and the parameter
    broadcast pass (handle, pointer, count, dtype code, stream) - driven under a gloo bootstrap with a recording stand-in for the library call,
    so that a typo in this path does not cost the hardware slot.  (The RCCL calls themselves run in tests/test_dp_gpu.py at world 1 and, on a
    multi-GPU box, at world 2.)"""
    script = tmp_path / "native_worker.py"
    script.write_text(NATIVE_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", str(script), ROOT]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and r.stdout.count("OK") == 2, r.stdout[-2000:] + r.stderr[-3000:]


SHARDED_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from transformers import AudioFlamingo3Config
from audio_flamingo_amd import ops
from audio_flamingo_amd.arena import FusedAdamW, ShardedAdamW, comm_share
from audio_flamingo_amd.modeling import AudioFlamingo3ForConditionalGeneration as Mine
from audio_flamingo_amd.dp import DataParallelEngine
import tests.test_host_cpu as T

# There is no CPU kernel path: the AdamW LAUNCH is replaced by a recording stand-in that applies torch.optim.AdamW's update in fp32 (csrc/elementwise.hip
# adamw_elem restated) to exactly the slices it is handed - the same stand-in serves the replicated and the sharded optimizer, so what this test pins is
# the HOST logic: the partition, the state mapping, which slices are launched, the collective sequence and that un-owned gradient shares are never read.
launches = []
def adamw_stub(master, m, v, grad, param, *, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, max_blocks=0, gate=None, hyper=None):
    assert master.numel() == m.numel() == v.numel() == grad.numel() == param.numel() and master.dtype == torch.float32 and param.dtype == torch.bfloat16
    launches.append((param.data_ptr(), param.numel()))
    if gate is not None and int(gate[0]) == 0:
        return
    g = grad.float() * grad_scale
    bc1, bc2s = 1.0 - beta1 ** step, (1.0 - beta2 ** step) ** 0.5
    master.mul_(1.0 - lr * weight_decay)
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    master.sub_((lr / bc1) * (m / (v.sqrt() / bc2s + eps)))
    param.copy_(master.to(torch.bfloat16))
ops.adamw_step = adamw_stub

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
cfg = AudioFlamingo3Config(**T.TINY)

def run(form):
    os.environ["AFK_DP_FORM"] = form
    m = Mine(cfg, device="cpu", init_seed=5 + rank)       # different init per rank on purpose
    a = m.arena
    eng = DataParallelEngine(a)
    eng.poison_unowned = True
    eng.broadcast_parameters(0)
    opt = eng.make_optimizer(lr=1e-2, weight_decay=0.01)
    opt.sync_master()
    assert isinstance(opt, ShardedAdamW) == (form == "rs_adamw_ag")
    seq = []
    for step in range(3):
        a.zero_grad(); eng.begin_backward()
        g = torch.Generator().manual_seed(1000 * step + rank)
        local = (torch.randn(a.total, generator=g) * 0.1).to(torch.bfloat16)
        skip_audio = step == 2                       # all-text step on every rank: the audio buckets' gate reads 0
        for blk in reversed(a.order):
            if skip_audio and (blk.key.startswith("model.audio_tower") or blk.key.startswith("model.multi_modal_projector")):
                continue
            blk.grad.copy_(local[blk.offset: blk.offset + blk.numel].view(blk.shape)); a.grad_written(blk)
        eng.finish()
        launches.clear()
        opt.step(grad_scale=eng.grad_scale, gates=eng.bucket_gate, refresh_shadows=False)
        seq.append((a.params.clone(), list(launches)))
        assert bool(torch.isfinite(a.params.float()).all()), (form, step, "non-finite parameters")
    return m, opt, seq

m_rep, opt_rep, rep = run("rs_ag")
m_sh, opt_sh, sh = run("rs_adamw_ag")
a = m_sh.arena
for step, ((p_rep, l_rep), (p_sh, l_sh)) in enumerate(zip(rep, sh)):
    assert torch.equal(p_rep, p_sh), (step, "sharded AdamW + parameter all-gather != replicated AdamW", float((p_rep.float() - p_sh.float()).abs().max()))
    # every launch of the sharded optimizer lies inside this rank's share or the tail of its bucket; together they cover exactly the owned elements
    base = a.params.data_ptr()
    covered = 0
    for ptr, n in l_sh:
        lo = (ptr - base) // 2
        assert any(a0 <= lo and lo + n <= a1 for a0, a1, _ in opt_sh.owned), (step, lo, n)
        covered += n
    assert covered == opt_sh.state_numel, (covered, opt_sh.state_numel)
    assert sum(n for _, n in l_rep) == a.total
# 1 / world of the optimizer state (+ the replicated tails)
tails = sum((e - s) - comm_share(e - s, world) * world for s, e in (a.bucket_range(i) for i in range(len(a.bucket_names))))
assert opt_sh.state_numel == (a.total - tails) // world + tails, (opt_sh.state_numel, a.total, tails)
assert opt_sh.master.numel() == opt_sh.m.numel() == opt_sh.v.numel() == opt_sh.state_numel < 0.55 * opt_rep.master.numel()
# the fp32 master of the owned pieces tracks the parameters everybody holds after the all-gather
for a0, a1, off in opt_sh.owned:
    assert torch.equal(opt_sh.master[off: off + a1 - a0].to(torch.bfloat16), a.params[a0:a1])
# replicas identical
chk = [None] * world
dist.all_gather_object(chk, [float(a.params.float().sum()), float(a.params.float().abs().sum())])
assert all(c == chk[0] for c in chk), chk
# checkpoint / resume (ADVICE r05): the consolidated state of the sharded optimizer IS the replicated optimizer's state, bit for bit, on every rank ...
sd = opt_sh.state_dict()
assert sd["state"]["t"] == opt_rep.t == 3 and sd["layout"]["numel"] == a.total
for name in ("master", "m", "v"):
    assert sd["state"][name].shape == (a.total,) and torch.equal(sd["state"][name], getattr(opt_rep, name)), name
# ... and loads back into a fresh sharded optimizer of a differently initialised replica: compact state, step count and every bf16 parameter restored
os.environ["AFK_DP_FORM"] = "rs_adamw_ag"
m2 = Mine(cfg, device="cpu", init_seed=99 + rank)
e2 = DataParallelEngine(m2.arena)
o2 = e2.make_optimizer(lr=1e-2, weight_decay=0.01)
o2.load_state_dict(sd)
assert o2.t == 3 and torch.equal(o2.master, opt_sh.master) and torch.equal(o2.m, opt_sh.m) and torch.equal(o2.v, opt_sh.v)
assert torch.equal(m2.arena.params, a.params)
# the same checkpoint resumes a REPLICATED optimizer (other form / other world size): same layout as trainer.AfkAdamW saves
o3 = FusedAdamW(m2.arena, lr=1e-2, weight_decay=0.01)
o3.master.copy_(sd["state"]["master"]); o3.m.copy_(sd["state"]["m"]); o3.v.copy_(sd["state"]["v"])
assert torch.equal(o3.master, opt_rep.master)
dist.destroy_process_group()
print("OK", rank)
'''


@pytest.mark.parametrize("nproc", [2, 4, 8])
def test_sharded_adamw_equals_replicated_gloo_world2(tmp_path, nproc):
    """AFK_DP_FORM=rs_adamw_ag - reduce-scatter, AdamW on this rank's 1 / world share of every bucket (arena.ShardedAdamW), all-gather
    of the bf16 parameters - gives BIT-IDENTICAL parameters to the replicated path over three steps (incl. an all-text step whose audio buckets are gated
    off), with 1 / world of the fp32 state, launches confined to the owned slices and un-owned gradient shares left unreduced.  World 2 over gloo on CPU arenas;
    the AdamW launch is a recording torch stand-in (no CPU kernels exist) shared by both paths.  Reference hook point: ddp_comm_hooks/default_hooks.py:18-35."""
    script = tmp_path / "sharded_worker.py"
    script.write_text(SHARDED_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT)
    env.pop("AFK_DP_FORM", None)
    env["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(29547 + nproc), str(script), ROOT]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and r.stdout.count("OK") == nproc, r.stdout[-2000:] + r.stderr[-3000:]


def test_comm_share_matches_the_library():
    """the host-side partition (arena.comm_share) and the C ABI's (afk_comm_share) must agree for every size / world / dtype"""
    from audio_flamingo_amd import _lib
    from audio_flamingo_amd.arena import comm_share

    lib = _lib.load()
    for n in (0, 1, 63, 64, 127, 128, 1000, 4096 * 129 + 7, 233057792):
        for world in (1, 2, 3, 8):
            assert int(lib.afk_comm_share(n, world, 0)) == comm_share(n, world, 2), (n, world)
            assert int(lib.afk_comm_share(n, world, 1)) == comm_share(n, world, 4), (n, world)


def test_reference_rotary_buffers_survive_the_harness_dtype_changes():
    """round 5 finding: `module.to(torch.bfloat16)` rounds the reference's non-persistent rotary `inv_freq` buffer (modeling_qwen2.py:67-68) and a later
    `.float()` only widens the rounded values, while a checkpoint loaded with `from_pretrained(dtype=bf16)` keeps the fp32 buffer (`:87`: explicit fp32 arange) - the
    behaviour this repo implements.  At theta = 1e6 the rounded frequencies put position 1 000 off by radians; the parity harness therefore restores the fp32 values
    after every dtype change of the reference (tools/parity_fulldepth.restore_rope_buffers).  Pinned here on the CPU: the hazard exists and the restore removes it."""
    from transformers import AudioFlamingo3Config, AudioFlamingo3ForConditionalGeneration

    from tools.parity_fulldepth import restore_rope_buffers

    cfg = dict(TINY)
    cfg["text_config"] = dict(TINY["text_config"], rope_parameters=dict(rope_theta=1000000.0, rope_type="default"))
    ref = AudioFlamingo3ForConditionalGeneration(AudioFlamingo3Config(**cfg))
    rot = next(m for m in ref.modules() if hasattr(m, "inv_freq") and hasattr(m, "compute_default_rope_parameters"))
    exact = rot.inv_freq.clone()
    assert exact.dtype == torch.float32
    ref.to(torch.bfloat16).float()
    rounded = rot.inv_freq.float()
    assert not torch.equal(rounded, exact), "this transformers build no longer rounds the buffer: the restore below is then a no-op"
    angle_err = ((rounded - exact).abs() * 1000.0).max()          # radians at position 1 000
    assert float(angle_err) > 0.5, float(angle_err)
    restore_rope_buffers(ref)
    assert rot.inv_freq.dtype == torch.float32 and torch.equal(rot.inv_freq, exact) and torch.equal(rot.original_inv_freq, exact)
    restore_rope_buffers(ref.to(torch.bfloat16))                   # and in a bf16 model the buffer stays fp32 (what from_pretrained(dtype=bf16) leaves)
    assert rot.inv_freq.dtype == torch.float32 and torch.equal(rot.inv_freq, exact)


def test_sharded_optimizer_ownership_map_and_budget_plan_accounting():
    """host arithmetic of round 5: (i) ShardedAdamW's ownership map - own share + replicated tail per bucket, compact state offsets, `_pieces` of arbitrary
    arena ranges - for worlds 2 / 3 / 8 on the tiny arena (no device, no launches); (ii) the memory-budgeted checkpoint plan takes the smaller of
    "budget - allocated" and what the device can still give, and subtracts allocations that are still to come (lazy W^T shadows)"""
    import bench

    from audio_flamingo_amd.arena import ShardedAdamW, comm_share
    from audio_flamingo_amd.modeling import AudioFlamingo3ForConditionalGeneration as Mine

    m = Mine(_cfg(), device="cpu")
    a = m.arena

    class Eng:   # rank / world are all the constructor reads of the engine
        def __init__(self, rank, world):
            self.rank, self.world = rank, world

    for world in (2, 3, 8):
        owned_all = []
        for rank in range(world):
            o = ShardedAdamW.__new__(ShardedAdamW)
            o.arena, o.engine, o.rank, o.world = a, Eng(rank, world), rank, world
            o.owned, off = [], 0
            for i in range(len(a.bucket_names)):
                s, e = a.bucket_range(i)
                share = comm_share(e - s, world)
                for a0, a1 in ((s + rank * share, s + (rank + 1) * share), (s + share * world, e)):
                    if a1 > a0:
                        o.owned.append((a0, a1, off))
                        off += a1 - a0
            assert o.owned == sorted(o.owned) and all(x[1] <= y[0] for x, y in zip(o.owned, o.owned[1:]))      # ascending, disjoint
            assert [p[2] for p in o.owned] == [sum(q[1] - q[0] for q in o.owned[:k]) for k in range(len(o.owned))]   # compact state
            s0, e0 = a.bucket_range(3)
            pieces = o._pieces(s0 + 5, e0 - 3)
            assert all(s0 + 5 <= lo < hi <= e0 - 3 for lo, hi, _ in pieces)
            assert sum(hi - lo for lo, hi, _ in o._pieces(0, a.total)) == off
            owned_all.append(o.owned)
        cover = torch.zeros(a.total, dtype=torch.int32)
        for rank, owned in enumerate(owned_all):
            for a0, a1, _ in owned:
                cover[a0:a1] += 1
        tails = torch.zeros(a.total, dtype=torch.bool)
        for i in range(len(a.bucket_names)):
            s, e = a.bucket_range(i)
            tails[s + comm_share(e - s, world) * world: e] = True
        assert bool((cover[~tails] == 1).all()) and bool((cover[tails] == world).all())   # every element owned once, tails by every rank

    g = Mine.__new__(Mine)   # geometry only (as test_checkpoint_planner_...)
    cfg = bench.af3_7b_config()
    ac, tc = cfg.audio_config, cfg.text_config
    g.config, g.enc_layers, g.dec_layers, g.max_pos, g.enc_heads = cfg, ac.num_hidden_layers, tc.num_hidden_layers, ac.max_source_positions, ac.num_attention_heads
    g.Hq, g.Hkv, g.D = tc.num_attention_heads, tc.num_key_value_heads, tc.hidden_size // tc.num_attention_heads
    g.ckpt_policy, g.ckpt_budget_bytes = "budget", None
    total, resident = 288 * 2 ** 30, int(139.5 * 2 ** 30)
    roomy = g.plan_checkpointing(20, 15274, resident, total, usable_bytes=140 * 2 ** 30)
    tight = g.plan_checkpointing(20, 15274, resident, total, usable_bytes=100 * 2 ** 30)       # another process holds 40 GiB of the device
    pending = g.plan_checkpointing(20, 15274, resident, total, usable_bytes=140 * 2 ** 30, pending_bytes=20 * 2 ** 30)
    assert roomy["dec"] < tight["dec"] and roomy["dec"] < pending["dec"], (roomy, tight, pending)
    assert g.plan_checkpointing(20, 15274, resident, total)["dec"] <= roomy["dec"]                # unknown device state: the budget alone


def _attn_qblock_host(L, Hq, Hkv, B, nz, causal):
    """host restatement of csrc/attention_lds.hip::attn_qblock (the XCD-aware block -> (sample, head, query block) map of the forward and dQ kernels, round 6)"""
    n, gsz, T = Hq * B, Hq // Hkv, Hq * B * nz
    if not causal:
        if T % 8 == 0:
            idx = (L & 7) * (T >> 3) + (L >> 3)
            item, lvl = divmod(idx, nz)
        else:
            lvl, item = divmod(L, n)
    else:
        k = 1 if n % 8 == 0 else 2 if n % 4 == 0 else 4 if n % 2 == 0 else 8
        full, c = nz // k, (k * n) >> 3
        if L < full * k * n:
            sup, r = divmod(L >> 3, c)
            idx = (L & 7) * c + r
            group, rem = divmod(idx, k * gsz)
            li, hh = divmod(rem, gsz)
            lvl = sup * k + li
            item = (group // Hkv) * Hq + (group % Hkv) * gsz + hh
        else:
            lvl, item = divmod(L - full * k * n, n)
            lvl += full * k
    b, h = divmod(item, Hq)
    return b, h, (nz - 1 - lvl if causal else lvl)


def test_attention_xcd_block_map_is_a_bijection_and_keeps_gqa_groups_on_one_xcd():
    """round 6: the forward / dQ kernels run on a 1-D grid whose block L lands on XCD L % 8; the map must (i) visit every (sample, head, query block) exactly once
    for every head layout / batch / block count - including counts that do not divide by 8 -, (ii) at the AF3 decoder shape give every XCD whole GQA groups (4 of the
    32 (sample, kv head) pairs: 2 MB of K / V in a 4 MB L2) and dispatch the long causal blocks first, (iii) with B = 1 give an XCD pair one kv head.  The device
    function is exercised by tests/test_ops_gpu.py::test_attention_schedules_bit_equal (same bits with the map on and off); this is its host restatement."""
    for Hq, Hkv in ((28, 4), (20, 20), (8, 2), (4, 4), (4, 2), (7, 1), (3, 3)):
        for B in (1, 2, 3, 5, 8):
            for nz in (1, 2, 3, 8, 12, 61):
                for causal in (True, False):
                    T = Hq * B * nz
                    seen = {_attn_qblock_host(L, Hq, Hkv, B, nz, causal) for L in range(T)}
                    assert len(seen) == T and all(0 <= b < B and 0 <= h < Hq and 0 <= z < nz for b, h, z in seen), (Hq, Hkv, B, nz, causal)
    per_xcd, order = {}, []
    for L in range(28 * 8 * 8):
        b, h, z = _attn_qblock_host(L, 28, 4, 8, 8, True)
        per_xcd.setdefault(L & 7, set()).add((b, h // 7))
        order.append(z)
    assert all(len(v) == 4 for v in per_xcd.values()) and len(set().union(*per_xcd.values())) == 32
    assert order == sorted(order, reverse=True)                      # level-major, longest (z = 7) first
    per_xcd = {}
    for L in range(28 * 60):                                         # B = 1, an even number of levels: levels pair up, one kv head per XCD pair
        b, h, z = _attn_qblock_host(L, 28, 4, 1, 60, True)
        per_xcd.setdefault(L & 7, set()).add(h // 7)
    assert all(v == {x // 2} for x, v in per_xcd.items()), per_xcd
    heads = {}
    for L in range(20 * 8 * 12):                                     # encoder (not causal): the 12 query blocks of a head sit on ONE XCD, back to back
        b, h, z = _attn_qblock_host(L, 20, 20, 8, 12, False)
        heads.setdefault((b, h), set()).add(L & 7)
    assert all(len(v) == 1 for v in heads.values())
