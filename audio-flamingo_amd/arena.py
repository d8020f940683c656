"""Flat parameter / gradient arenas for one model replica (sized for 288 GB of HBM3E per GPU).

Every trainable tensor of the model is a view into ONE bf16 parameter buffer; its gradient is the matching view of
ONE bf16 gradient buffer.  This gives
  * the optimizer a single fused launch over contiguous memory (afk_adamw_step, 28 B/param of traffic),
  * the data-parallel engine contiguous per-layer buckets to all-reduce (no flatten/unflatten copies),
  * wgrad kernels a fixed destination: they write (or accumulate) straight into the arena, autograd never
    allocates a weight-gradient tensor.
Blocks are laid out in FORWARD order and grouped into buckets (one per transformer layer); backward produces
them in reverse, so bucket i becomes reducible as soon as layer i's backward has been enqueued.

``Shadow`` tensors are the K-major copies of the GEMM weights that bring dgrad to the same NT kernel as the
forward (W^T for Linear, tap-major permutations for the conv stem).  They are refreshed after each optimizer
step (or lazily when a parameter's ``_version`` changes, so external optimizers keep working).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import torch

from . import ops

ALIGN = 64  # elements (128 B): keeps every block 16-byte aligned for the LDS-DMA GEMM


@dataclass
class Block:
    key: str
    shape: tuple
    bucket: int
    decay: bool
    offset: int = 0
    numel: int = 0
    data: Optional[torch.Tensor] = None
    grad: Optional[torch.Tensor] = None
    fresh: bool = True  # no gradient written since the last zero_grad -> next wgrad overwrites instead of accumulating
    shadow: Optional[torch.Tensor] = None
    shadow_kind: Optional[str] = None  # "T" (W^T), "conv" (tap-major + its transpose)
    shadow_aux: Optional[torch.Tensor] = None
    shadow_version: int = -1


class Arena:
    def __init__(self, device):
        self.device = torch.device(device)
        self.blocks: Dict[str, Block] = {}
        self.order: List[Block] = []
        self.bucket_names: List[str] = []
        self.params: Optional[torch.Tensor] = None
        self.grads: Optional[torch.Tensor] = None
        self.total = 0
        self.step_counter = 0  # bumped by our optimizer (raw-pointer writes do not touch tensor._version)
        self._bucket_ranges: List[tuple] = []
        self._bucket_pending: List[int] = []
        self._bucket_sizes: List[int] = []
        self.on_bucket_ready: Optional[Callable[[int], None]] = None
        # optional second compute stream for the weight-gradient branch of every Linear backward (functional.linear_bwd):
        # dgrad and wgrad of a layer are independent given dY, so running them on two streams lets one GEMM's blocks fill
        # the partially empty last wave of the other (448-tile GEMMs leave 25 % of the CUs idle in their second round)
        self.wgrad_stream: Optional[torch.cuda.Stream] = None
        self.thin_blocks = 0  # grid cap for the transposes issued on the wgrad stream (0 = full grid)
        # W^T shadows on demand only: set when dgrad runs on the NN kernel (functional.BWD_FORM == "direct"), where most of them
        # are never read; the eager per-bucket refresh after the optimizer then skips them and arena.shadow() rebuilds a stale one
        self.lazy_T_shadows = False

    # ------------------------------------------------------------------ layout
    def new_bucket(self, name: str) -> int:
        self.bucket_names.append(name)
        return len(self.bucket_names) - 1

    def add(self, key: str, shape, bucket: int, decay: bool = True, shadow: Optional[str] = None) -> Block:
        assert self.params is None, "arena already finalized"
        b = Block(key=key, shape=tuple(shape), bucket=bucket, decay=decay, shadow_kind=shadow)
        b.numel = 1
        for s in b.shape:
            b.numel *= s
        self.blocks[key] = b
        self.order.append(b)
        return b

    def finalize(self):
        off = 0
        nb = len(self.bucket_names)
        starts, ends = [None] * nb, [0] * nb
        for b in self.order:
            b.offset = off
            off += (b.numel + ALIGN - 1) // ALIGN * ALIGN
            if starts[b.bucket] is None:
                starts[b.bucket] = b.offset
            ends[b.bucket] = off
        self.total = off
        self.params = torch.zeros(off, device=self.device, dtype=torch.bfloat16)
        self.grads = torch.zeros(off, device=self.device, dtype=torch.bfloat16)
        for b in self.order:
            b.data = self.params[b.offset: b.offset + b.numel].view(b.shape)
            b.grad = self.grads[b.offset: b.offset + b.numel].view(b.shape)
        self._bucket_ranges = [(starts[i] or 0, ends[i]) for i in range(nb)]
        self._bucket_sizes = [sum(1 for b in self.order if b.bucket == i) for i in range(nb)]
        self._bucket_pending = list(self._bucket_sizes)

    def __getitem__(self, key: str) -> Block:
        return self.blocks[key]

    def bucket_range(self, i: int):
        return self._bucket_ranges[i]

    def bucket_grads(self, i: int) -> torch.Tensor:
        s, e = self._bucket_ranges[i]
        return self.grads[s:e]

    # ------------------------------------------------------------------ gradient bookkeeping
    def zero_grad(self, memset: bool = False):
        """Mark every block fresh (first wgrad of the step overwrites).  memset=True also clears the buffer
        (needed only for tensors that receive sparse updates, e.g. embed_tokens: handled by its Function)."""
        for b in self.order:
            b.fresh = True
        self._bucket_pending = list(self._bucket_sizes)
        if memset:
            self.grads.zero_()

    def grad_written(self, blk: Block):
        """called by the wgrad sites after enqueueing the kernels that finish blk.grad for this backward"""
        if blk.fresh:
            blk.fresh = False
        self._bucket_pending[blk.bucket] -= 1
        if self._bucket_pending[blk.bucket] == 0 and self.on_bucket_ready is not None:
            self.on_bucket_ready(blk.bucket)

    def begin_backward(self):
        self._bucket_pending = list(self._bucket_sizes)

    def enable_wgrad_stream(self, on: bool = True):
        self.wgrad_stream = torch.cuda.Stream(device=self.device) if on and self.device.type == "cuda" else None

    def join_streams(self):
        """make the current stream wait for everything enqueued on the wgrad stream (call before reading .grad)"""
        if self.wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)

    def ready_events(self):
        """events that together cover every gradient kernel enqueued so far (compute stream + wgrad stream)"""
        evs = [torch.cuda.Event()]
        evs[0].record(torch.cuda.current_stream())
        if self.wgrad_stream is not None:
            e = torch.cuda.Event()
            e.record(self.wgrad_stream)
            evs.append(e)
        return evs

    # ------------------------------------------------------------------ shadows
    def _refresh_one(self, b: Block):
        if b.shadow_kind == "T":
            w2 = b.data.reshape(b.shape[0], -1)
            if b.shadow is None:
                b.shadow = torch.empty((w2.shape[1], ops.pad64(w2.shape[0])), device=self.device, dtype=torch.bfloat16)
            ops.transpose(w2, out=b.shadow, rpad=b.shadow.shape[1])
        elif b.shadow_kind == "conv":
            co, ci, _ = b.shape
            b.shadow = ops.conv_weight_to_gemm(b.data, out=b.shadow)  # [Co, 3*Ci]
            if b.shadow_aux is None:
                b.shadow_aux = torch.empty((3 * ci, ops.pad64(co)), device=self.device, dtype=torch.bfloat16)
            ops.transpose(b.shadow, out=b.shadow_aux, rpad=b.shadow_aux.shape[1])
        b.shadow_version = self._version_of(b)

    def _version_of(self, b: Block) -> int:
        return self.params._version * 1000003 + self.step_counter

    def refresh_bucket_shadows(self, i: int):
        for b in self.order:
            if b.bucket == i and b.shadow_kind is not None and not (self.lazy_T_shadows and b.shadow_kind == "T"):
                self._refresh_one(b)

    def refresh_shadows(self, force: bool = True):
        for b in self.order:
            if self.lazy_T_shadows and b.shadow_kind == "T":
                continue
            if b.shadow_kind is not None and (force or b.shadow_version != self._version_of(b)):
                self._refresh_one(b)

    def shadow(self, key: str) -> torch.Tensor:
        b = self.blocks[key]
        if b.shadow is None or b.shadow_version != self._version_of(b):
            self._refresh_one(b)
        return b.shadow

    def shadow_aux(self, key: str) -> torch.Tensor:
        self.shadow(key)
        return self.blocks[key].shadow_aux

    # ------------------------------------------------------------------ init (oracle: _init_weights, normal(0, initializer_range))
    def init_normal_(self, std: float, seed: int = 0):
        g = torch.Generator(device=self.device)
        g.manual_seed(seed)
        chunk = 1 << 28
        for s in range(0, self.total, chunk):
            e = min(self.total, s + chunk)
            self.params[s:e] = (torch.randn(e - s, device=self.device, dtype=torch.float32, generator=g) * std).to(torch.bfloat16)


class FusedAdamW:
    """AdamW over the arena: bf16 params + fp32 master/m/v.  One launch per decay class."""

    def __init__(self, arena: Arena, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.arena = arena
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.master = arena.params.float()
        self.m = torch.zeros_like(self.master)
        self.v = torch.zeros_like(self.master)
        self.t = 0
        # contiguous segments sharing a decay setting
        self.segments = []
        for b in arena.order:
            end = b.offset + (b.numel + ALIGN - 1) // ALIGN * ALIGN
            wd = weight_decay if b.decay else 0.0
            if self.segments and self.segments[-1][2] == wd and self.segments[-1][1] == b.offset:
                self.segments[-1][1] = end
            else:
                self.segments.append([b.offset, end, wd])

        # per-bucket segments (for optimizer-in-backward overlap)
        self.bucket_segments = [[] for _ in arena.bucket_names]
        for blk in arena.order:
            end = blk.offset + (blk.numel + ALIGN - 1) // ALIGN * ALIGN
            wd = weight_decay if blk.decay else 0.0
            segs = self.bucket_segments[blk.bucket]
            if segs and segs[-1][2] == wd and segs[-1][1] == blk.offset:
                segs[-1][1] = end
            else:
                segs.append([blk.offset, end, wd])

    def _launch(self, s, e, wd, grad_scale, max_blocks=0):
        a = self.arena
        ops.adamw_step(self.master[s:e], self.m[s:e], self.v[s:e], a.grads[s:e], a.params[s:e], lr=self.lr,
                       beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, weight_decay=wd, step=self.t,
                       grad_scale=grad_scale, max_blocks=max_blocks)

    def begin_step(self):
        """overlapped mode: advance the step count once, then step_bucket() per bucket as its gradients complete"""
        self.t += 1

    def step_bucket(self, i: int, grad_scale: float = 1.0, max_blocks: int = 0):
        for s, e, wd in self.bucket_segments[i]:
            self._launch(s, e, wd, grad_scale, max_blocks)

    def end_step(self):
        self.arena.step_counter += 1
        for b in self.arena.order:  # shadows were refreshed bucket by bucket (lazy W^T shadows stay stale until used)
            if b.shadow_kind is not None and not (self.arena.lazy_T_shadows and b.shadow_kind == "T"):
                b.shadow_version = self.arena._version_of(b)

    def step(self, grad_scale: float = 1.0, refresh_shadows: bool = True):
        self.t += 1
        a = self.arena
        a.join_streams()
        for s, e, wd in self.segments:
            self._launch(s, e, wd, grad_scale)
        a.step_counter += 1
        if refresh_shadows:
            a.refresh_shadows(force=True)

    def zero_grad(self):
        self.arena.zero_grad()
