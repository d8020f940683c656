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
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from .arena import Arena


class DataParallelEngine:
    def __init__(self, arena: Arena, process_group=None, overlap: bool = True):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.arena = arena
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.cuda = arena.grads.is_cuda
        self.overlap = overlap and self.cuda
        self.comm_stream = torch.cuda.Stream(device=arena.device) if self.cuda else None
        self._works: List = []
        self._done = [False] * len(arena.bucket_names)
        self.enabled = True  # set False inside a no_sync() region (gradient accumulation micro-steps)
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
        chunk = 1 << 28
        p = self.arena.params
        for s in range(0, p.numel(), chunk):
            dist.broadcast(p[s: s + chunk], src=src, group=self.pg)
        self.arena.step_counter += 1

    # ------------------------------------------------------------------ bucket exchange
    def begin_backward(self):
        self._done = [False] * len(self.arena.bucket_names)
        self._works = []
        self.arena.begin_backward()

    def _reduce(self, i: int):
        buf = self.arena.bucket_grads(i)
        if buf.numel() == 0:
            return
        if self.cuda:
            evs = self.arena.ready_events()
            with torch.cuda.stream(self.comm_stream):
                for ev in evs:
                    self.comm_stream.wait_event(ev)
                w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self._works.append(w)
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
        for i in range(len(self._done)):
            if not self._done[i] or (self.cuda and not self.overlap):
                self._done[i] = True
                self._reduce(i)
        for w in self._works:
            w.wait()  # on the nccl backend this makes the CURRENT stream wait; it does not block the host
        self._works = []
        self.arena.join_streams()
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

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


class BackwardOverlap:
    """Runs, per gradient bucket and as soon as backward has produced it: [RCCL sum-all-reduce] -> fused AdamW on that
    bucket -> refresh of its W^T shadows, all on a side HIP stream while backward continues on the compute stream.

    Backward is MFMA-bound and leaves HBM mostly idle; AdamW is pure HBM traffic (28 B/param, 231 GB/step for AF3-7B =
    40 ms at 5.9 TB/s) and the all-reduce is xGMI traffic, so both hide almost completely in the GEMM shadow.  The
    arithmetic of the step is unchanged: every parameter is updated once, with its final (reduced) gradient.  Not usable
    with global-norm gradient clipping or on non-final gradient-accumulation micro-steps (call the plain path there).
    """

    def __init__(self, arena: Arena, optimizer, engine: Optional["DataParallelEngine"] = None):
        self.arena, self.opt, self.engine = arena, optimizer, engine
        self.side = torch.cuda.Stream(device=arena.device)
        self._done: List[bool] = []
        self.grad_scale = engine.grad_scale if engine is not None else 1.0
        self.thin_blocks = 256  # optimizer launches of one block per CU so that they co-reside with the GEMM workgroups

    def begin_step(self):
        self.opt.begin_step()
        self._done = [False] * len(self.arena.bucket_names)
        self.arena.begin_backward()
        self.arena.on_bucket_ready = self._ready

    def _ready(self, i: int):
        if self._done[i]:
            return
        self._done[i] = True
        evs = self.arena.ready_events()
        with torch.cuda.stream(self.side):
            for ev in evs:
                self.side.wait_event(ev)
            if self.engine is not None and self.engine.world > 1:
                buf = self.arena.bucket_grads(i)
                if buf.numel():
                    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.engine.pg)  # ordered on the side stream
            self.opt.step_bucket(i, self.grad_scale, self.thin_blocks)
            self.arena.refresh_bucket_shadows(i)

    def finish(self):
        for i in range(len(self._done)):
            self._ready(i)
        self.arena.join_streams()
        torch.cuda.current_stream().wait_stream(self.side)
        self.opt.end_step()
        self.arena.on_bucket_ready = self.engine._on_bucket_ready if self.engine is not None else None
