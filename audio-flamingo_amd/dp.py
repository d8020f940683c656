"""Data-parallel gradient exchange: one process per GPU, RCCL over xGMI.

The oracle's path is torch DDP (TORCH/nn/parallel/distributed.py:662-666, 25 MiB buckets, C++ Reducer).  Here the
gradient arena (arena.py) already holds each transformer layer's gradients contiguously, so a bucket is just a
slice of it: no flatten/copy, no per-parameter hooks.  The wgrad sites call ``arena.grad_written(block)``; when the
last block of a bucket has been enqueued the engine
    1. records a HIP event on the compute stream,
    2. makes the communication stream wait on it,
    3. enqueues one sum-all-reduce of that slice on the communication stream (RCCL via torch.distributed "nccl").
Backward keeps running on the compute stream, so the exchange of layer i overlaps the backward of layer i-1.
Buckets are whole layers (466 MB for a decoder layer) rather than 25 MiB: xGMI is a point-to-point mesh and large
messages amortise RCCL's per-collective launch across all 7 links.  Averaging is folded into the
optimizer (grad_scale = 1/world), so the reduction itself is a pure sum.

Collective order is rank-independent by construction.  Backward completes the buckets in exactly reverse layout order
(head, dec27 .. dec0, embed, enc_out, enc31 .. enc0, stem); a rank whose batch skipped part of the model (a text-only
batch never runs the audio tower) has reduced a PREFIX of that sequence when backward ends, and ``finish()`` issues the
rest in the same reverse order with the unwritten gradient slices cleared first - so every rank issues the same
collectives in the same order with the same sizes (DDP raises for this case unless find_unused_parameters is set,
TORCH/nn/parallel/distributed.py:1442 ff.).  One more tiny collective closes the step: a MAX over per-bucket "touched"
flags; the optimizer launches are gated on it ON THE DEVICE, so a bucket that no rank touched keeps its parameters and
moments exactly as torch.optim does for ``grad is None`` (DDP's find_unused_parameters bitmap, distributed.py:1533 ff.).
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist

from .arena import Arena, comm_share


def _flags_on_device(flags, device) -> torch.Tensor:
    """int32 device vector of 0/1 host flags, written by FILL kernels on the current stream (no host-to-device copy: a pageable H2D copy
    cannot be recorded into a HIP graph, fill launches can - the step's flags are constants of a captured static-shape step)"""
    t = torch.zeros(len(flags), device=device, dtype=torch.int32)
    if all(flags):
        t.fill_(1)
    else:
        i, n = 0, len(flags)
        while i < n:          # one fill per run of ones
            if flags[i]:
                j = i
                while j < n and flags[j]:
                    j += 1
                t[i:j].fill_(1)
                i = j
            else:
                i += 1
    return t


class NativeComm:
    """The RCCL communicator behind the C ABI (include/afk.h afk_comm_*): what a non-Python host of libafk.so would use, and - with
    AFK_DP_COMM=native - what DataParallelEngine uses instead of torch.distributed's collectives.  One communicator per process / GPU."""

    _DT = {torch.bfloat16: 0, torch.float32: 1, torch.int32: 2}

    def __init__(self, rank: int, world: int, uid: bytes):
        import ctypes

        from . import _lib

        self.rank, self.world = rank, world
        h = ctypes.c_void_p()
        _lib.call("afk_comm_init", rank, world, uid, ctypes.addressof(h))
        self.handle = h

    @staticmethod
    def unique_id() -> bytes:
        import ctypes

        from . import _lib

        buf = ctypes.create_string_buffer(128)
        _lib.call("afk_comm_unique_id", ctypes.addressof(buf))
        return buf.raw

    @classmethod
    def from_process_group(cls, process_group=None):
        """bootstrap over an existing torch.distributed group: rank 0 creates the id, everybody receives it"""
        rank, world = dist.get_rank(process_group), dist.get_world_size(process_group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=process_group)
        return cls(rank, world, box[0])

    def allreduce_(self, t: torch.Tensor, *, form: str = "rs_ag", op_max: bool = False):
        """in-place sum (or max) over the ranks, enqueued on the CURRENT stream.  form "rs_ag": reduce-scatter + all-gather (every xGMI link
        of the mesh busy), "allreduce": one ncclAllReduce"""
        from . import _lib
        from .ops import _stream

        assert t.is_cuda and t.is_contiguous() and t.dtype in self._DT
        if form == "rs_ag" and not op_max:
            _lib.call("afk_reduce_scatter_allgather_bucket", self.handle, t.data_ptr(), t.numel(), self._DT[t.dtype], _stream())
        else:
            _lib.call("afk_allreduce_bucket", self.handle, t.data_ptr(), t.numel(), self._DT[t.dtype], int(op_max), _stream())
        return t

    def reduce_scatter_(self, t: torch.Tensor):
        """in place: this rank's share (afk_comm_share) and the replicated tail of `t` hold the SUM over the ranks afterwards"""
        from . import _lib
        from .ops import _stream

        assert t.is_cuda and t.is_contiguous() and t.dtype in self._DT
        _lib.call("afk_reduce_scatter_bucket", self.handle, t.data_ptr(), t.numel(), self._DT[t.dtype], _stream())
        return t

    def allgather_(self, t: torch.Tensor):
        """in place: every rank's own share of `t` is distributed to all ranks (the tail is left alone)"""
        from . import _lib
        from .ops import _stream

        assert t.is_cuda and t.is_contiguous() and t.dtype in self._DT
        _lib.call("afk_allgather_bucket", self.handle, t.data_ptr(), t.numel(), self._DT[t.dtype], _stream())
        return t

    def broadcast_(self, t: torch.Tensor, root: int = 0):
        from . import _lib
        from .ops import _stream

        _lib.call("afk_comm_broadcast", self.handle, t.data_ptr(), t.numel(), self._DT[t.dtype], root, _stream())
        return t

    def close(self):
        from . import _lib

        if self.handle:
            _lib.call("afk_comm_destroy", self.handle)
            self.handle = None


class DataParallelEngine:
    _native_cache: dict = {}   # one RCCL communicator of the C ABI per (process group) for the life of the process: engines come and go (bench.py's form probe)

    def __init__(self, arena: Arena, process_group=None, overlap: bool = True, comm: Optional[str] = None, form: Optional[str] = None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.arena = arena
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.cuda = arena.grads.is_cuda
        self.backend = dist.get_backend(process_group)
        # device buffers over a host-side backend (gloo): stage through the host.  Only for correctness runs of the multi-rank path on a
        # box without one GPU per rank (tests/test_dp_gpu.py: two ranks sharing one MI355X); the production backend is "nccl" = RCCL.
        self.staged = self.cuda and self.backend != "nccl"
        # which library issues the collectives: "torch" = torch.distributed (c10d -> RCCL), "native" = libafk.so's own communicator
        # (afk_comm_* C ABI, reduce-scatter + all-gather form); env AFK_DP_COMM / AFK_DP_FORM.  Same arithmetic, same order.
        import os as _os

        self.comm_kind = comm or _os.environ.get("AFK_DP_COMM", "torch")
        self.form = form or _os.environ.get("AFK_DP_FORM", "rs_ag")
        # AFK_DP_FORM=rs_adamw_ag (opt-in, round 5): the optimizer is SHARDED over the ranks - per bucket reduce-scatter(grads) -> AdamW on this rank's
        # share (arena.ShardedAdamW) -> all-gather(bf16 params); with either communicator library.  After reduce_bucket_() only this rank's share and
        # the replicated tail of a gradient bucket hold reduced values.
        self.sharded = self.form == "rs_adamw_ag"
        self._gate_vec: Optional[torch.Tensor] = None
        self.poison_unowned = False   # tests: where the exchange is emulated by an all-reduce (gloo), put the LOCAL values back into the shares this rank does not own
        self.native = None
        if self.comm_kind == "native" and self.cuda and not self.staged:
            ck = id(process_group) if process_group is not None else 0
            if ck not in DataParallelEngine._native_cache:
                DataParallelEngine._native_cache[ck] = NativeComm.from_process_group(process_group)
            self.native = DataParallelEngine._native_cache[ck]
        self.overlap = overlap and self.cuda
        self.comm_stream = torch.cuda.Stream(device=arena.device) if self.cuda else None
        self._works: List = []
        self._done = [False] * len(arena.bucket_names)
        self.issued: List[int] = []  # bucket indices in the order their collectives were issued this step (tests compare ranks)
        self.bucket_gate: Optional[torch.Tensor] = None  # device int32 [n_buckets] after finish(): 1 where ANY rank touched the bucket
        self.enabled = True  # set False inside a no_sync() region (gradient accumulation micro-steps)
        # world == 1 normally issues no collective at all.  force_collectives (bench.py --force-dp, env AFK_DP_FORCE=1) issues them anyway - a
        # sum over one rank is the identity - so that the WHOLE multi-rank code path (side-stream ordering, RCCL launches, the touched-flag MAX,
        # device-gated AdamW, RCCL inside a HIP-graph capture) can be executed and checked bit for bit on a 1-GPU box
        self.force_collectives = _os.environ.get("AFK_DP_FORCE", "0") == "1"
        arena.on_bucket_ready = self._on_bucket_ready
        self._native_bf16 = True
        if not self.cuda:
            # gloo: reduce through fp32 when the backend lacks bf16 support
            try:
                t = torch.zeros(2, dtype=torch.bfloat16)
                dist.all_reduce(t, group=process_group)
            except Exception:
                self._native_bf16 = False

    # ------------------------------------------------------------------ parameter broadcast (DDP ctor, distributed.py:1012)
    def broadcast_parameters(self, src: int = 0):
        """rank `src`'s replica state to every rank: the parameter arena and the frozen tensors beside it (arena.extra_state)"""
        chunk = 1 << 28
        for p in [self.arena.params] + [t.view(-1) for t in self.arena.extra_state]:
            for s in range(0, p.numel(), chunk):
                if self.staged:
                    h = p[s: s + chunk].float().cpu()
                    dist.broadcast(h, src=src, group=self.pg)
                    p[s: s + chunk].copy_(h.to(p.device))
                else:
                    dist.broadcast(p[s: s + chunk], src=src, group=self.pg)
        self.arena.step_counter += 1

    # ------------------------------------------------------------------ bucket exchange
    def begin_backward(self):
        self._done = [False] * len(self.arena.bucket_names)
        self._works = []
        self.issued = []
        self.bucket_gate = None
        self.arena.begin_backward()

    def allreduce_sum_(self, buf: torch.Tensor):
        """sum-all-reduce of a device slice, ordered on the CURRENT stream (RCCL), or staged through the host (gloo: synchronous)"""
        if self.native is not None:
            self.native.allreduce_(buf, form=self.form)
            return
        if not self.staged:
            if not self.cuda and not self._native_bf16:   # host arenas over a gloo build without bf16 reductions: through fp32
                f = buf.float()
                dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.pg)
                buf.copy_(f)
                return
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg)
            return
        for ev in self.arena.ready_events():
            ev.synchronize()
        torch.cuda.current_stream().synchronize()
        f = buf.float().cpu()
        dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.pg)
        buf.copy_(f.to(buf.device))

    # ------------------------------------------------------------------ sharded form: reduce-scatter / all-gather halves
    def make_optimizer(self, **kw):
        """the optimizer that matches this engine's exchange form: arena.ShardedAdamW for rs_adamw_ag, arena.FusedAdamW otherwise"""
        from .arena import FusedAdamW, ShardedAdamW

        return ShardedAdamW(self.arena, self, **kw) if self.sharded else FusedAdamW(self.arena, **kw)

    def _shares(self, i: int):
        s, e = self.arena.bucket_range(i)
        return s, e, comm_share(e - s, self.world)

    def reduce_bucket_(self, i: int):
        """SUM of gradient bucket i over the ranks, ordered on the CURRENT stream.  Replicated forms: the whole bucket is reduced everywhere.
        Sharded form: reduce-scatter - afterwards only this rank's share and the replicated tail hold reduced values."""
        buf = self.arena.bucket_grads(i)
        if not self.sharded:
            return self.allreduce_sum_(buf)
        s, e, share = self._shares(i)
        n, r, w = e - s, self.rank, self.world
        if self.native is not None:
            self.native.reduce_scatter_(buf)
        elif self.cuda and not self.staged:
            if share:
                dist.reduce_scatter_tensor(buf[r * share:(r + 1) * share], buf[: share * w], op=dist.ReduceOp.SUM, group=self.pg)
            if share * w < n:
                dist.all_reduce(buf[share * w:], op=dist.ReduceOp.SUM, group=self.pg)
        else:
            # gloo (host staging / CPU arenas: correctness runs only) has no reduce-scatter: all-reduce; with poison_unowned (tests) the shares this
            # rank does not own then get their LOCAL values back - what an in-place ncclReduceScatter leaves there - so that an update which read
            # an un-owned share would differ from the replicated path (NaN would do too, but never-written slices such as the padding between
            # blocks or the k third of the fused encoder qkv bias are updated by the launches as well and must stay finite)
            local = buf.clone() if (self.poison_unowned and share) else None
            if self.cuda:
                self.allreduce_sum_(buf)
            elif self._native_bf16:
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg)
            else:
                f = buf.float()
                dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.pg)
                buf.copy_(f)
            if local is not None:
                if r > 0:
                    buf[: r * share].copy_(local[: r * share])
                if r + 1 < w:
                    buf[(r + 1) * share: share * w].copy_(local[(r + 1) * share: share * w])

    def allgather_params_(self, i: int):
        """sharded form, after this rank's AdamW launches of bucket i (same stream): every rank's share of the bucket's bf16 parameters to all ranks"""
        if not self.sharded:
            return
        s, e, share = self._shares(i)
        if share == 0 or (self.world == 1 and not self.force_collectives):
            return
        # `.data`: same storage, its OWN version counter - the parameter views are saved tensors of backward nodes that have not run yet (this call
        # sits inside backward), and an in-place torch op on the arena itself would invalidate them for autograd; the AdamW kernels write through
        # raw pointers for the same reason
        self._allgather_shares_(self.arena.params.data[s:e], share)

    def _allgather_shares_(self, p: torch.Tensor, share: int):
        """in place on the CURRENT stream: rank r's [r * share, (r + 1) * share) of `p` to every rank"""
        r, w = self.rank, self.world
        if self.native is not None:
            self.native.allgather_(p)
        elif self.cuda and not self.staged:
            dist.all_gather_into_tensor(p[: share * w], p[r * share:(r + 1) * share], group=self.pg)
        else:
            mine = p[r * share:(r + 1) * share].detach().cpu().contiguous().view(torch.int32)   # bit pattern (shares are multiples of 128 bytes): gloo moves bytes
            outs = [torch.empty_like(mine) for _ in range(w)]
            dist.all_gather(outs, mine, group=self.pg)
            p[: share * w].copy_(torch.cat(outs).view(torch.bfloat16).to(p.device))

    def exchange_bucket_flag_(self, i: int, touched: bool) -> torch.Tensor:
        """sharded form inside BackwardOverlap: MAX over the ranks of "this rank produced a gradient for bucket i", on the CURRENT stream, right behind the
        bucket's reduce-scatter -> device int32[1] that gates the bucket's AdamW launches.  Why per bucket: the parameter all-gather that follows the
        update is a collective too, so every rank must issue  RS_i, flag_i, AG_i  in the same order for every bucket - a rank that skipped part of the
        model (text-only batch) issues them from finish(), the others from inside backward, and the one closing MAX over all flags of the replicated forms
        would sit at different places of the two sequences."""
        if self._gate_vec is None:
            self._gate_vec = torch.zeros(len(self.arena.bucket_names), device=self.arena.device, dtype=torch.int32)
        g = self._gate_vec[i:i + 1]
        if self.staged or not self.cuda:
            h = torch.tensor([int(touched)], dtype=torch.int32)
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.pg)
            g.copy_(h.to(g.device))
        else:
            g.fill_(int(touched))    # a fill launch: capturable into a HIP graph (the flag is a constant of a static-shape step)
            if self.native is not None:
                self.native.allreduce_(g, op_max=True)
            else:
                dist.all_reduce(g, op=dist.ReduceOp.MAX, group=self.pg)
        return g

    def allreduce_small_sum_(self, t: torch.Tensor):
        """SUM over the ranks of a small fp32 device vector on the CURRENT stream (the sharded optimizer's partial sums of squares)"""
        if self.world == 1 and not self.force_collectives:
            return t
        if self.native is not None:
            self.native.allreduce_(t, form="allreduce")
        elif self.staged:
            h = t.detach().cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.pg)
            t.copy_(h.to(t.device))
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
        return t

    def _reduce(self, i: int):
        buf = self.arena.bucket_grads(i)
        if buf.numel() == 0:
            return
        self.arena.zero_unwritten(i)  # zero_grad() only flips flags: slices nobody wrote this step still hold the last step's values
        self.issued.append(i)
        if self.sharded:
            if self.staged or not self.cuda:
                self.reduce_bucket_(i)
            else:
                evs = self.arena.ready_events()
                with torch.cuda.stream(self.comm_stream):
                    for ev in evs:
                        self.comm_stream.wait_event(ev)
                    self.reduce_bucket_(i)   # enqueued on (c10d: ordered behind and ahead of) the communication stream; finish() joins it
        elif self.staged:
            self.allreduce_sum_(buf)
        elif self.cuda:
            evs = self.arena.ready_events()
            with torch.cuda.stream(self.comm_stream):
                for ev in evs:
                    self.comm_stream.wait_event(ev)
                if self.native is not None:
                    self.native.allreduce_(buf, form=self.form)  # enqueued on the communication stream; finish() joins it
                else:
                    self._works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        elif self._native_bf16:
            self._works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        else:
            f = buf.float()
            dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.pg)
            buf.copy_(f)

    def _on_bucket_ready(self, i: int):
        if not self.enabled or self._done[i]:
            return
        self._done[i] = True
        if self.overlap or not self.cuda:
            self._reduce(i)

    def finish(self):
        """after loss.backward(): exchange whatever is still pending, then order the compute stream after the comm stream"""
        if not self.enabled:
            return
        touched = [int(self.arena.bucket_touched(i)) for i in range(len(self._done))]
        for i in reversed(range(len(self._done))):  # REVERSE layout order = the order backward would have produced them (see module doc)
            if not self._done[i] or (self.cuda and not self.overlap):
                self._done[i] = True
                self._reduce(i)
        self.bucket_gate = self._exchange_touched(touched)
        for w in self._works:
            w.wait()  # on the nccl backend this makes the CURRENT stream wait; it does not block the host
        self._works = []
        self.arena.join_streams()
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        for b in self.arena.order:  # every slice now holds a reduced gradient (zeros where nobody contributed)
            b.fresh = False

    def _exchange_touched(self, touched) -> torch.Tensor:
        """MAX over ranks of the per-bucket "a gradient was produced here" flags; always the LAST collective of a step"""
        if self.staged:
            t = torch.tensor(touched, dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.pg)
            t = t.to(self.arena.device)
        elif self.cuda:
            with torch.cuda.stream(self.comm_stream):
                t = _flags_on_device(touched, self.arena.device)
                if self.native is not None:
                    self.native.allreduce_(t, op_max=True)
                else:
                    self._works.append(dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.pg, async_op=True))
            t.record_stream(torch.cuda.current_stream())
        else:
            t = torch.tensor(touched, dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.pg)
        return t

    # ------------------------------------------------------------------ pre-flight (bench.py, N > 1)
    def preflight(self) -> dict:
        """One round of every collective the step uses, on a real gradient bucket, with KNOWN answers - run once before the first step so that a broken
        communicator shows up as a named failure in seconds instead of as a hang or a diverged replica mid-run:
            all-reduce(SUM) of a whole layer bucket (the replicated forms) -> every element = world (world + 1) / 2;
            reduce-scatter of it (the sharded form) -> this rank's share and the tail hold that sum; all-gather of per-rank markers -> share q = q + 1;
            the touched-flag MAX; replica checksum (parameters identical on every rank after broadcast_parameters).
        -> {"ok", "failed": [...], "<leg>_ms", "allreduce_busbw_gbs", ...}.  Clobbers the gradient arena (re-zeroed at the end), never the parameters."""
        import time

        a, w, r = self.arena, self.world, self.rank
        sizes = [a.bucket_range(i)[1] - a.bucket_range(i)[0] for i in range(len(a.bucket_names))]
        i = max((k for k in range(len(sizes)) if 2 * sizes[k] <= (1 << 30)), key=lambda k: sizes[k], default=max(range(len(sizes)), key=lambda k: sizes[k]))
        buf, n = a.bucket_grads(i), sizes[i]
        share = comm_share(n, w)
        want = w * (w + 1) / 2.0
        out = {"bucket": a.bucket_names[i], "bytes": 2 * n, "world": w, "comm": self.comm_kind if self.native is not None else ("gloo-staged" if self.staged else "torch"),
               "failed": []}

        def sync():
            if self.cuda:
                torch.cuda.synchronize()

        def timed(name, fn):
            sync()
            t0 = time.perf_counter()
            fn()
            sync()
            out[name + "_ms"] = round(1e3 * (time.perf_counter() - t0), 2)

        def all_equal(t, v):
            return t.numel() == 0 or (float(t.float().min()) == v and float(t.float().max()) == v)

        buf.fill_(r + 1)
        timed("allreduce_first", lambda: self.allreduce_sum_(buf))     # carries the lazy communicator / channel setup
        if not all_equal(buf, want):
            out["failed"].append("allreduce")
        buf.fill_(r + 1)
        timed("allreduce", lambda: self.allreduce_sum_(buf))
        out["allreduce_busbw_gbs"] = round(2.0 * (w - 1) / max(w, 1) * 2 * n / max(out["allreduce_ms"] * 1e-3, 1e-9) / 1e9, 1)
        was = self.sharded
        try:
            self.sharded = True
            buf.fill_(r + 1)
            timed("reduce_scatter", lambda: self.reduce_bucket_(i))
            mine = torch.cat([buf[r * share:(r + 1) * share], buf[share * w:]])
            if not all_equal(mine, want):
                out["failed"].append("reduce_scatter")
            if share and (w > 1 or self.force_collectives):
                buf[: share * w].zero_()
                buf[r * share:(r + 1) * share].fill_(r + 1)
                timed("all_gather", lambda: self._allgather_shares_(buf, share))
                if not all(all_equal(buf[q * share:(q + 1) * share], float(q + 1)) for q in range(w)):
                    out["failed"].append("all_gather")
        finally:
            self.sharded = was
        flags = [1 if (k % w) == r else 0 for k in range(len(sizes))]
        if self.cuda and not self.staged:
            g = _flags_on_device(flags, a.device)
            if self.native is not None:
                self.native.allreduce_(g, op_max=True)
            else:
                dist.all_reduce(g, op=dist.ReduceOp.MAX, group=self.pg)
        else:
            g = torch.tensor(flags, dtype=torch.int32)
            dist.all_reduce(g, op=dist.ReduceOp.MAX, group=self.pg)
        sync()
        if int(g.min()) != 1 and len(sizes) >= w:
            out["failed"].append("flag_max")
        p32 = a.params.float()
        cdev = a.device if (self.cuda and not self.staged) else torch.device("cpu")
        chk = torch.stack([p32.sum().double(), p32.abs().sum().double()]).to(cdev)
        allc = [torch.zeros_like(chk) for _ in range(w)]
        dist.all_gather(allc, chk, group=self.pg)
        if not all(bool(torch.equal(c, allc[0])) for c in allc):
            out["failed"].append("replica_checksum")
        del p32
        a.grads.zero_()
        sync()
        # every rank must reach the same verdict: a leg that failed anywhere failed
        bad = torch.tensor([len(out["failed"])], dtype=torch.int32, device=a.device if (self.cuda and not self.staged) else "cpu")
        dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=self.pg)
        out["ok"] = int(bad.item()) == 0
        return out

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    class _NoSync:
        def __init__(self, eng):
            self.eng = eng

        def __enter__(self):
            self.eng.enabled = False

        def __exit__(self, *a):
            self.eng.enabled = True

    def no_sync(self):
        """gradient-accumulation micro-steps: gradients accumulate locally, nothing is exchanged (DDP.no_sync, distributed.py:1442)"""
        return DataParallelEngine._NoSync(self)


_PROBE_SKIP_ADAMW = os.environ.get("AFK_PROBE_SKIP_ADAMW", "0") == "1"   # bench.py marks a run with it INVALID


class BackwardOverlap:
    """Runs, per gradient bucket and as soon as backward has produced it: [RCCL sum-all-reduce] -> fused AdamW on that
    bucket -> refresh of its W^T shadows, all on a side HIP stream while backward continues on the compute stream.

    Backward is MFMA-bound and leaves HBM mostly idle; AdamW is pure HBM traffic (28 B/param, 231 GB/step for AF3-7B =
    40 ms at 5.9 TB/s) and the all-reduce is xGMI traffic, so both hide almost completely in the GEMM shadow.  The
    arithmetic of the step is unchanged: every parameter is updated once, with its final (reduced) gradient.  Not usable
    on non-final gradient-accumulation micro-steps (call the plain path there).
    With global-norm clipping (``optimizer.clip_norm``) the coefficient needs every gradient, so only the all-reduces and the partial
    sums of squares run inside backward; the AdamW launches (which apply the coefficient) and the shadow refreshes follow in finish().
    """

    def __init__(self, arena: Arena, optimizer, engine: Optional["DataParallelEngine"] = None):
        self.arena, self.opt, self.engine = arena, optimizer, engine
        from .streams import make_stream

        self.side = make_stream(arena.device, "side")
        self._done: List[bool] = []
        self.grad_scale = engine.grad_scale if engine is not None else 1.0
        self.thin_blocks = 768  # optimizer launches of three blocks per CU: thin enough to co-reside with the GEMM workgroups (a full grid locks them out), and 1.5 ms per step faster
                                # than one block per CU under the low-priority side stream (round 6, call 24: 256 / 512 / 768 / 1024 / 2048 blocks = 390.9 / 389.3 / 388.7 / 389.3 / 389.7 ms)
        # measure_tail (eager steps only - captured events carry no timestamps): HIP-event pairs around the wait for the side stream at the end of every
        # step = how long the compute stream sat idle behind the last backward kernel waiting for [exchange ->] AdamW -> shadow refresh of the last buckets
        self.measure_tail = False
        self.tail_events: List[tuple] = []

    def begin_step(self):
        self.opt.begin_step()
        self._done = [False] * len(self.arena.bucket_names)
        self._clip_pending = []  # (bucket, gate, written_only) whose AdamW waits for the clip coefficient
        if self.engine is not None:
            self.engine.issued = []
        self.arena.begin_backward()
        self.arena.on_bucket_ready = self._ready

    def _multi(self) -> bool:
        return self.engine is not None and (self.engine.world > 1 or self.engine.force_collectives)

    def _ready(self, i: int, gate=None, written_only: bool = False):
        if self._done[i]:
            return
        self._done[i] = True
        if self._multi():
            self.arena.zero_unwritten(i)  # on the compute stream, ahead of the events below
        evs = self.arena.ready_events()
        with torch.cuda.stream(self.side):
            for ev in evs:
                self.side.wait_event(ev)
            if self._multi():
                buf = self.arena.bucket_grads(i)
                if buf.numel():
                    self.engine.issued.append(i)
                    self.engine.reduce_bucket_(i)  # ordered on the side stream (all-reduce, or reduce-scatter in the sharded form)
                if self.engine.sharded:            # RS_i, flag_i, [AdamW on this rank's share], AG_i: the same sequence on every rank (exchange_bucket_flag_)
                    gate, written_only = self.engine.exchange_bucket_flag_(i, True), False
            self._step_or_defer(i, gate, written_only)

    def exposed_tail_ms(self, last: int = 0):
        """mean over the recorded steps (the last `last` of them; 0 = all) of the exposed tail; the events must have completed (synchronize first)"""
        ev = self.tail_events[-last:] if last else self.tail_events
        return None if not ev else sum(a.elapsed_time(b) for a, b in ev) / len(ev)

    def _step_or_defer(self, i, gate, written_only):
        """on the side stream, behind bucket i's reduction: AdamW + shadow refresh - or, with clipping, its share of the gradient norm"""
        if getattr(self.opt, "clip_norm", None):
            self.opt.add_sumsq(i, gate=gate, written_only=written_only)
            self._clip_pending.append((i, gate, written_only))
            return
        if _PROBE_SKIP_ADAMW:   # timing probe (wrong training): what the step costs without its optimizer launches
            return
        fused = self.opt.step_bucket(i, self.grad_scale, self.thin_blocks, gate=gate, written_only=written_only)
        self.arena.refresh_bucket_shadows(i, skip=fused)   # W^T shadows the optimizer launch wrote itself are skipped

    def _flush_clipped(self):
        if not self._clip_pending:
            return
        with torch.cuda.stream(self.side):
            self.opt.set_clip_coef(self.grad_scale)
            for i, gate, written_only in self._clip_pending:
                fused = self.opt.step_bucket(i, self.grad_scale, 0, gate=gate, written_only=written_only)  # backward is over: full-width launches
                self.arena.refresh_bucket_shadows(i, skip=fused)
        self._clip_pending = []

    def _exchange_flags_and_step(self, touched, deferred):
        """replicated forms: ONE MAX over all per-bucket flags behind the last reduction, then the gated optimizer launches of the deferred buckets"""
        eng = self.engine
        with torch.cuda.stream(self.side):
            if eng.staged:
                gate = torch.tensor(touched, dtype=torch.int32)
                dist.all_reduce(gate, op=dist.ReduceOp.MAX, group=eng.pg)
                gate = gate.to(self.arena.device)
            else:
                gate = _flags_on_device(touched, self.arena.device)
                if eng.native is not None:
                    eng.native.allreduce_(gate, op_max=True)
                else:
                    dist.all_reduce(gate, op=dist.ReduceOp.MAX, group=eng.pg)
            for i in deferred:
                self._step_or_defer(i, gate[i:i + 1], False)
        return gate

    def finish(self):
        """buckets backward never completed (their part of the model did not run this step, e.g. the audio tower on a text-only batch):
        single process -> the optimizer skips what received no gradient (torch: grad is None); data parallel -> they are reduced in the
        same reverse order every rank uses, and their AdamW launches are gated on the all-rank "touched" flags (dp.py module doc)."""
        n = len(self._done)
        pending = [i for i in reversed(range(n)) if not self._done[i]]
        if self._multi():
            eng = self.engine
            touched = [int(self.arena.bucket_touched(i)) for i in range(n)]
            deferred = []
            for i in pending:  # reductions first (same order on every rank), flags after them, gated optimizer launches last
                self._done[i] = True
                self.arena.zero_unwritten(i)
                evs = self.arena.ready_events()
                with torch.cuda.stream(self.side):
                    for ev in evs:
                        self.side.wait_event(ev)
                    buf = self.arena.bucket_grads(i)
                    if buf.numel():
                        eng.issued.append(i)
                        eng.reduce_bucket_(i)
                    if eng.sharded:   # the bucket's own flag exchange, its gated update and its parameter all-gather follow at once (see _ready)
                        self._step_or_defer(i, eng.exchange_bucket_flag_(i, bool(touched[i])), False)
                if not eng.sharded:
                    deferred.append(i)
            if eng.sharded:
                gate = eng._gate_vec if eng._gate_vec is not None else _flags_on_device(touched, self.arena.device)
            else:
                gate = self._exchange_flags_and_step(touched, deferred)
            gate.record_stream(torch.cuda.current_stream())
            eng.bucket_gate = gate
        else:
            for i in pending:
                if self.arena.bucket_touched(i):
                    self._ready(i, written_only=True)
                else:
                    self._done[i] = True
        self._flush_clipped()
        self.arena.join_streams()
        if self.measure_tail:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            torch.cuda.current_stream().wait_stream(self.side)
            e1.record()
            self.tail_events.append((e0, e1))
        else:
            torch.cuda.current_stream().wait_stream(self.side)
        self.opt.end_step()
        self.arena.on_bucket_ready = self.engine._on_bucket_ready if self.engine is not None else None
