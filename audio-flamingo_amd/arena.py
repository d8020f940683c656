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

import os

import torch

from . import ops
from ._lib import AfkError

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
    shadow_lazy: bool = False  # the eager per-step refresh skips this block; arena.shadow() rebuilds it on demand (the training step never reads it)


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
        self.presums = {}   # one hand-over slot {for: bias key, ptr: data_ptr of the gradient tensor, row: bf16 column sums}: left by a gradient's producer for the bias gradient of its consumer (functional.py)
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
        # replica state that lives OUTSIDE the trainable buffer (frozen tensors: the encoder's embed_positions table): the data-parallel
        # engine broadcasts it together with the parameters so that replicas start identical (DDP broadcasts buffers too)
        self.extra_state: List[torch.Tensor] = []

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
        self._bucket_blocks = [[b for b in self.order if b.bucket == i] for i in range(nb)]
        self._bucket_sizes = [len(bl) for bl in self._bucket_blocks]
        self._bucket_pending = list(self._bucket_sizes)

    def __getitem__(self, key: str) -> Block:
        return self.blocks[key]

    def bucket_range(self, i: int):
        return self._bucket_ranges[i]

    def bucket_grads(self, i: int) -> torch.Tensor:
        s, e = self._bucket_ranges[i]
        return self.grads[s:e]

    def bucket_blocks(self, i: int) -> List[Block]:
        return self._bucket_blocks[i]

    def bucket_touched(self, i: int) -> bool:
        """has any block of bucket i received a gradient since the last zero_grad()?  (a text-only batch never runs the audio tower:
        those buckets stay untouched and - as torch does for ``grad is None`` - must not be fed to the optimizer with stale contents)"""
        return any(not b.fresh for b in self._bucket_blocks[i])

    def zero_unwritten(self, i: int) -> int:
        """clear the gradient slices of bucket i that no kernel has written since zero_grad() (zero_grad only flips the `fresh`
        flags, the buffer still holds the previous step's values); -> number of blocks cleared.  Used before a bucket is handed
        to a collective / optimizer launch that covers the whole slice."""
        n = 0
        for b in self._bucket_blocks[i]:
            if b.fresh:
                b.grad.zero_()
                n += 1
        return n

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
        self.presums.clear()

    def enable_wgrad_stream(self, on: bool = True):
        from .streams import make_stream   # lowest queue priority: the wgrad GEMMs fill what the critical path leaves (streams.py)

        self.wgrad_stream = make_stream(self.device, "wgrad") if on and self.device.type == "cuda" else None

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

    def version(self) -> int:
        """changes whenever a parameter may have changed: in-place tensor writes (load_state_dict, torch ops) and our optimizer's raw-pointer steps"""
        return self.params._version * 1000003 + self.step_counter

    def refresh_bucket_shadows(self, i: int, skip=None):
        """skip: blocks whose shadow the optimizer launch itself just wrote (FusedAdamW with FUSE_SHADOW)"""
        for b in self._bucket_blocks[i]:
            if b.shadow_kind is not None and not b.shadow_lazy and not (self.lazy_T_shadows and b.shadow_kind == "T") and not (skip is not None and b.key in skip):
                self._refresh_one(b)

    def refresh_shadows(self, force: bool = True, skip=None):
        for b in self.order:
            if (self.lazy_T_shadows and b.shadow_kind == "T") or b.shadow_lazy or (skip is not None and b.key in skip):
                continue
            if b.shadow_kind is not None and (force or b.shadow_version != self._version_of(b)):
                self._refresh_one(b)

    def ensure_T_shadow(self, b: Block) -> torch.Tensor:
        """the [K, pad64(N)] buffer of a "T" block (allocated on first use; contents are whatever was last written)"""
        if b.shadow is None:
            n, k = b.shape[0], b.numel // b.shape[0]
            b.shadow = torch.zeros((k, ops.pad64(n)), device=self.device, dtype=torch.bfloat16)
        return b.shadow

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
        self._synced = (arena.params._version, arena.step_counter)
        # step-dependent scalars (lr, bias corrections) live on the device: every AdamW launch reads them there, so a captured HIP graph
        # of the whole training step replays correctly after advance() has refreshed them
        self.hyper = torch.zeros(4, device=arena.device, dtype=torch.float32) if arena.device.type == "cuda" else None
        self._in_capture = False  # graphs.GraphedTrainStep: the capture pass records launches only
        # global-norm gradient clipping (torch.nn.utils.clip_grad_norm_; HF Trainer default max_grad_norm = 1.0): the norm is one streaming
        # pass over the flat gradient arena, the coefficient rides in hyper[3] and is applied inside the AdamW launches
        self.clip_norm: Optional[float] = None
        if self.hyper is not None:
            self._sumsq = torch.zeros(len(arena.bucket_names), device=arena.device, dtype=torch.float32)  # one slot per gradient bucket
            self.grad_norm = torch.zeros(1, device=arena.device, dtype=torch.float32)  # norm of the last clipped step (device scalar)
            self._sumsq_ws = torch.empty(ops.sumsq_workspace_floats(), device=arena.device, dtype=torch.float32)
        # OPTIONAL, OFF by default: the W^T shadow of a 2-D GEMM weight (the dgrad operand) written by the optimizer launch itself
        # (afk_adamw_step_t) instead of a transpose pass per weight - 15.4 GB fewer reads and ~245 fewer launches per AF3-7B step on paper.
        # Built, bit-identical (tests/test_model_gpu.py::test_adamw_fused_transposed_shadow) and MEASURED SLOWER on the full step, same box:
        # 440-442 vs 429-431 ms with thin launches of 256 blocks, 429 with 512, 434 vs 426 ms on the serial schedule - the 64 x 64-tiled
        # launch (two barriers and an LDS transpose per tile, 128-byte bf16 row segments, <= 64 VGPRs so that it fits beside the GEMM waves)
        # streams the 28 B/param of optimizer state slower than the flat kernel by more than the transposes cost.  AFK_ADAMW_FUSE_SHADOW=1 enables it.
        self.fuse_shadow = os.environ.get("AFK_ADAMW_FUSE_SHADOW", "0") == "1" and arena.device.type == "cuda"
        # launch plans: ("flat", start, end, wd) = contiguous run of blocks sharing a decay setting; ("T", block, wd) = one fused weight.
        # Built on first use and rebuilt when the arena's shadow policy changes (bench.py sets lazy_T_shadows after constructing the optimizer)
        self._plan_key = None

    @property
    def segments(self):
        self._ensure_plans()
        return self._segments

    @property
    def bucket_segments(self):
        self._ensure_plans()
        return self._bucket_segments

    def _ensure_plans(self):
        # the set of lazily-shadowed blocks is part of the key: linear_bwd / LMHeadLossFn flip Block.shadow_lazy at run time (a weight whose dgrad
        # reads W itself needs no W^T shadow, and a fused launch would allocate and rewrite one for nothing)
        lazy = tuple(b.key for b in self.arena.order if b.shadow_lazy) if self.fuse_shadow else ()
        key = (self.fuse_shadow, self.arena.lazy_T_shadows, self.weight_decay, lazy)
        if key != self._plan_key:
            a = self.arena
            self._segments = self._plan(a.order)
            self._bucket_segments = [self._plan(a.bucket_blocks(i)) for i in range(len(a.bucket_names))]
            self._plan_key = key

    def _fusable(self, b) -> bool:
        return (self.fuse_shadow and b.shadow_kind == "T" and not b.shadow_lazy and not self.arena.lazy_T_shadows and len(b.shape) == 2 and b.shape[0] % 64 == 0
                and b.shape[1] % 64 == 0 and b.offset % 8 == 0)

    def _plan(self, blocks):
        plan = []
        for b in blocks:
            end = b.offset + (b.numel + ALIGN - 1) // ALIGN * ALIGN
            wd = self.weight_decay if b.decay else 0.0
            if self._fusable(b):
                plan.append(["T", b, wd])
            elif plan and plan[-1][0] == "flat" and plan[-1][3] == wd and plan[-1][2] == b.offset:
                plan[-1][2] = end
            else:
                plan.append(["flat", b.offset, end, wd])
        return plan

    def _fused_keys(self, plan):
        return {op[1].key for op in plan if op[0] == "T"}

    def _exec(self, plan, grad_scale, max_blocks=0, gate=None):
        a = self.arena
        for op in plan:
            if op[0] == "flat":
                self._launch(op[1], op[2], op[3], grad_scale, max_blocks, gate)
            else:
                b, wd = op[1], op[2]
                s, e = b.offset, b.offset + b.numel
                ops.adamw_step_t(self.master[s:e], self.m[s:e], self.v[s:e], a.grads[s:e], a.params[s:e], a.ensure_T_shadow(b), b.shape[0], b.shape[1],
                                 lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, weight_decay=wd, step=self.t, grad_scale=grad_scale,
                                 max_blocks=max_blocks, gate=gate, hyper=self.hyper)

    def _launch(self, s, e, wd, grad_scale, max_blocks=0, gate=None):
        a = self.arena
        ops.adamw_step(self.master[s:e], self.m[s:e], self.v[s:e], a.grads[s:e], a.params[s:e], lr=self.lr,
                       beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, weight_decay=wd, step=self.t,
                       grad_scale=grad_scale, max_blocks=max_blocks, gate=gate, hyper=self.hyper)

    def advance(self):
        """t += 1 and publish (lr, 1 - beta1^t, sqrt(1 - beta2^t)) to the device.  Called by begin_step() / step(); a HIP-graph replay of
        the step calls it directly (the captured AdamW launches read the device copy)."""
        if self._in_capture:
            return
        self.t += 1
        if self.hyper is not None:
            ops.set_f32(self.hyper, [self.lr, 1.0 - self.betas[0] ** self.t, (1.0 - self.betas[1] ** self.t) ** 0.5, 1.0])
            if self.clip_norm:
                self._sumsq.zero_()

    # ------------------------------------------------------------------ global-norm clipping
    def add_sumsq(self, i: int, gate=None, written_only: bool = True):
        """sum(g^2) of bucket i's gradients (written_only: of the blocks written since zero_grad()) into the bucket's slot of the step's
        norm.  One slot per bucket, folded in index order by set_clip_coef(): the plain step and the overlapped schedule (which fills
        the slots in backward order, on a side stream) arrive at the same bits.  gate: device int32[1], dropped on the device when 0"""
        a = self.arena
        blocks = [b for b in a.bucket_blocks(i) if not (written_only and b.fresh)]
        for s, e, _ in self._runs(blocks):
            ops.sumsq_(a.grads[s:e], self._sumsq[i:i + 1], gate=gate, ws=self._sumsq_ws)

    def set_clip_coef(self, grad_scale: float = 1.0):
        """hyper[3] = min(1, clip_norm / (grad_scale * sqrt(sum g^2) + 1e-6)): the AdamW launches that follow scale every gradient by it"""
        ops.clip_coef_(self._sumsq, self.hyper[3:4], max_norm=self.clip_norm, scale=grad_scale, norm_out=self.grad_norm)

    # ------------------------------------------------------------------ master <-> working copy
    def sync_master(self):
        """re-read the fp32 master from the bf16 parameters (call after anything that rewrote them: load_state_dict, broadcast)"""
        self.master.copy_(self.arena.params)
        self._mark_synced()

    def _mark_synced(self):
        self._synced = (self.arena.params._version, self.arena.step_counter)

    def _check_master(self):
        """parameters rewritten behind the optimizer's back (model.load_state_dict, a broadcast, an in-place init) would be overwritten
        by `old master + update` on the next step: detect it (tensor version / arena step counter) and re-read the master first."""
        if self._synced != (self.arena.params._version, self.arena.step_counter):
            self.sync_master()

    def _runs(self, blocks):
        runs = []
        for b in blocks:
            end = b.offset + (b.numel + ALIGN - 1) // ALIGN * ALIGN
            wd = self.weight_decay if b.decay else 0.0
            if runs and runs[-1][2] == wd and runs[-1][1] == b.offset:
                runs[-1][1] = end
            else:
                runs.append([b.offset, end, wd])
        return runs

    def begin_step(self):
        """overlapped mode: advance the step count once, then step_bucket() per bucket as its gradients complete"""
        self._check_master()
        self.advance()

    def step_bucket(self, i: int, grad_scale: float = 1.0, max_blocks: int = 0, gate=None, written_only: bool = False):
        """AdamW on bucket i.  gate: device int32 - the launches do nothing when it reads 0 (data parallel: "did ANY rank touch this
        bucket", known only on the device).  written_only: cover just the blocks written since zero_grad() (torch skips ``grad is None``)."""
        plan = self._plan([b for b in self.arena.bucket_blocks(i) if not b.fresh]) if written_only else self.bucket_segments[i]
        self._exec(plan, grad_scale, max_blocks, gate)
        return self._fused_keys(plan)

    def end_step(self):
        self.arena.step_counter += 1
        self._mark_synced()
        for b in self.arena.order:  # shadows were refreshed bucket by bucket (lazy W^T shadows stay stale until used)
            if b.shadow_kind is not None and not b.shadow_lazy and not (self.arena.lazy_T_shadows and b.shadow_kind == "T"):
                b.shadow_version = self.arena._version_of(b)

    def step(self, grad_scale: float = 1.0, refresh_shadows: bool = True, gates=None):
        """one optimizer step over every block that received a gradient since zero_grad() (blocks whose backward did not run this
        step - e.g. the audio tower on a text-only batch - are skipped, as torch.optim skips ``grad is None``).
        gates (data parallel, from DataParallelEngine.finish()): device int32 [n_buckets], 1 where any rank touched the bucket -
        every rank then launches every bucket and the device decides."""
        self._check_master()
        self.advance()
        a = self.arena
        a.join_streams()
        if self.clip_norm:
            for i in range(len(a.bucket_names)):
                self.add_sumsq(i, gate=gates[i:i + 1] if gates is not None else None, written_only=gates is None)
            self.set_clip_coef(grad_scale)
        fused = set()
        if gates is not None:
            for i in range(len(a.bucket_names)):
                self._exec(self.bucket_segments[i], grad_scale, gate=gates[i:i + 1])
                fused |= self._fused_keys(self.bucket_segments[i])
        else:
            plan = self.segments if all(not b.fresh for b in a.order) else self._plan([b for b in a.order if not b.fresh])
            self._exec(plan, grad_scale)
            fused = self._fused_keys(plan)
        a.step_counter += 1
        self._mark_synced()
        if refresh_shadows:
            a.refresh_shadows(force=True, skip=fused)
            for k in fused:  # written by the optimizer launch itself
                a.blocks[k].shadow_version = a._version_of(a.blocks[k])

    def zero_grad(self):
        self.arena.zero_grad()


def comm_share(n: int, world: int, itemsize: int = 2) -> int:
    """elements of a bucket of n that each rank owns: (n / world) rounded down to a multiple of 128 bytes - include/afk.h afk_comm_share
    (tests/test_host_cpu.py holds the two to each other).  Rank r owns [r * share, (r + 1) * share); [share * world, n) is a replicated tail."""
    if n <= 0 or world <= 0:
        return 0
    align = 128 // itemsize
    return (n // world) // align * align


class ShardedAdamW(FusedAdamW):
    """FusedAdamW with the optimizer SHARDED over the data-parallel ranks (AFK_DP_FORM=rs_adamw_ag).

    Per gradient bucket (= transformer layer) the step is   reduce-scatter(grads) -> AdamW on THIS rank's share -> all-gather(bf16 params)
    instead of   all-reduce(grads) -> AdamW on the whole bucket   on every rank.  Same bytes on the wire (a reduce-scatter + an all-gather IS an
    all-reduce), but every GPU streams 1 / world of the 28 B/param optimizer traffic (AF3-7B: 231 GB -> 29 GB per step at 8 ranks) and holds 1 / world of
    the fp32 master / m / v (99.2 GB -> 12.4 GB).  The oracle's DDP replicates the optimizer (TORCH/nn/parallel/distributed.py:662-666, 828-834); the hook
    point it exposes for this is the communication hook (ddp_comm_hooks/default_hooks.py:18-35).

    AdamW is elementwise, so the parameters are BIT-IDENTICAL to the replicated path whenever the reduced gradient values are (same reduce-scatter as the
    replicated rs_ag form; at world 2 any summation order gives the same bits).  The tail of a bucket that does not divide (afk_comm_share: shares are
    multiples of 128 bytes) is all-reduced and updated by every rank - replicated state of < world * 64 elements per bucket.

    GRADIENT VALIDITY: after a sharded reduce only this rank's share and the replicated tail of `arena.grads` hold reduced values - the shares of the other
    ranks still hold this rank's LOCAL, unreduced gradients.  Nothing but this optimizer may read arena.grads in this form (a logging hook that wants a
    gradient norm reads `grad_norm`, which set_clip_coef() assembles from every rank's share).  engine.poison_unowned (tests) makes a wrong reader visible.

    State lives in ONE compact fp32 array per moment: for every bucket, this rank's share followed by the bucket's tail.  `engine` (dp.DataParallelEngine)
    supplies rank / world and the collectives; after this rank's launches of a bucket it all-gathers the bucket's parameters on the same stream.
    Global-norm clipping: every rank sums the squares of its own shares (the tail on rank 0 only), one extra SUM all-reduce of the per-bucket partial
    sums closes the norm - identical on every rank, NOT bit-identical to the replicated path's norm (other summation order)."""

    def __init__(self, arena: Arena, engine, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.arena = arena
        self.engine = engine
        self.rank, self.world = int(engine.rank), int(engine.world)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        # owned intervals of the arena, ascending: (arena start, arena end, state offset)
        self.owned: List[tuple] = []
        off = 0
        for i in range(len(arena.bucket_names)):
            s, e = arena.bucket_range(i)
            share = comm_share(e - s, self.world)
            for a0, a1 in ((s + self.rank * share, s + (self.rank + 1) * share), (s + share * self.world, e)):
                if a1 > a0:
                    self.owned.append((a0, a1, off))
                    off += a1 - a0
        self.state_numel = off
        self.master = torch.empty(off, device=arena.device, dtype=torch.float32)
        self.m = torch.zeros(off, device=arena.device, dtype=torch.float32)
        self.v = torch.zeros(off, device=arena.device, dtype=torch.float32)
        self.t = 0
        self.hyper = torch.zeros(4, device=arena.device, dtype=torch.float32) if arena.device.type == "cuda" else None
        self._in_capture = False
        self.clip_norm: Optional[float] = None
        self._sumsq = torch.zeros(len(arena.bucket_names), device=arena.device, dtype=torch.float32)
        self.grad_norm = torch.zeros(1, device=arena.device, dtype=torch.float32)
        self._sumsq_ws = torch.empty(ops.sumsq_workspace_floats(), device=arena.device, dtype=torch.float32) if arena.device.type == "cuda" else None
        self.fuse_shadow = False   # the transposed-shadow launch needs whole weights; a share cuts through them
        self._plan_key = None
        self.sync_master()

    # ------------------------------------------------------------------ ownership
    def _pieces(self, s: int, e: int):
        """the parts of arena range [s, e) this rank owns -> (arena start, arena end, state start)"""
        out = []
        for a0, a1, off in self.owned:
            if a1 <= s:
                continue
            if a0 >= e:
                break
            lo, hi = max(a0, s), min(a1, e)
            if hi > lo:
                out.append((lo, hi, off + lo - a0))
        return out

    def state_bytes(self) -> int:
        return 12 * self.state_numel

    def sync_master(self):
        for a0, a1, off in self.owned:
            self.master[off: off + a1 - a0].copy_(self.arena.params[a0:a1])
        self._mark_synced()

    # ------------------------------------------------------------------ launches
    def _launch(self, s, e, wd, grad_scale, max_blocks=0, gate=None):
        a = self.arena
        for lo, hi, off in self._pieces(s, e):
            n = hi - lo
            ops.adamw_step(self.master[off: off + n], self.m[off: off + n], self.v[off: off + n], a.grads[lo:hi], a.params[lo:hi], lr=self.lr,
                           beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, weight_decay=wd, step=self.t,
                           grad_scale=grad_scale, max_blocks=max_blocks, gate=gate, hyper=self.hyper)

    def add_sumsq(self, i: int, gate=None, written_only: bool = True):
        """this rank's part of sum(g^2) of bucket i: its own share, and the replicated tail on rank 0 only (counted once in the all-rank sum).
        Data parallel: every slice of a reduced bucket holds a reduced gradient (zeros where nobody contributed).  Without collectives (one process,
        force_collectives off) nothing has cleared the blocks backward did not write: `written_only` then restricts the sum to the blocks written
        since zero_grad(), exactly the set step_bucket() updates (the norm and the update must see the same gradients)."""
        a = self.arena
        s, e = a.bucket_range(i)
        share = comm_share(e - s, self.world)
        tail0 = s + share * self.world
        multi = self.world > 1 or bool(getattr(self.engine, "force_collectives", False))
        if written_only and not multi:
            spans = [(rs, re_) for rs, re_, _ in self._runs([b for b in a.bucket_blocks(i) if not b.fresh])]
        else:
            spans = [(s, e)]
        for rs, re_ in spans:
            for lo, hi, _ in self._pieces(rs, re_):
                if lo >= tail0 and self.rank != 0:
                    continue
                ops.sumsq_(a.grads[lo:hi], self._sumsq[i:i + 1], gate=gate, ws=self._sumsq_ws)

    def set_clip_coef(self, grad_scale: float = 1.0):
        self.engine.allreduce_small_sum_(self._sumsq)   # per-bucket partial sums of squares over the ranks' shares -> the full gradient's
        super().set_clip_coef(grad_scale)

    def step_bucket(self, i: int, grad_scale: float = 1.0, max_blocks: int = 0, gate=None, written_only: bool = False):
        """AdamW on this rank's share (+ the tail) of bucket i, then the all-gather of the bucket's bf16 parameters - both on the CURRENT stream"""
        # written_only arises in single-process use only (a bucket backward touched partly); a reduced bucket is defined everywhere
        plan = self._plan([b for b in self.arena.bucket_blocks(i) if not b.fresh]) if written_only else self.bucket_segments[i]
        self._exec(plan, grad_scale, max_blocks, gate)
        self.engine.allgather_params_(i)
        return set()

    def step(self, grad_scale: float = 1.0, refresh_shadows: bool = True, gates=None):
        self._check_master()
        self.advance()
        a = self.arena
        a.join_streams()
        n = len(a.bucket_names)
        if self.clip_norm:
            for i in range(n):
                self.add_sumsq(i, gate=gates[i:i + 1] if gates is not None else None)
            self.set_clip_coef(grad_scale)
        for i in range(n):
            self.step_bucket(i, grad_scale, gate=gates[i:i + 1] if gates is not None else None, written_only=gates is None)
        a.step_counter += 1
        self._mark_synced()
        if refresh_shadows:
            a.refresh_shadows(force=True)

    # ------------------------------------------------------------------ checkpoint / resume (the sharded state must be saveable)
    def _consolidate(self, compact: torch.Tensor, to_cpu: bool) -> torch.Tensor:
        """one compact per-rank state array -> the FULL flat fp32 array in the arena's layout (what FusedAdamW holds and trainer.AfkAdamW saves), on every
        rank: each rank writes the pieces it owns into zeros (a bucket's replicated tail from rank 0 only) and a SUM over the ranks assembles them -
        x + 0 + ... + 0 is exact, so the result is bit-identical to the owners' values.  Chunked (256 Mi elements) so that a host-staged backend works."""
        a = self.arena
        full = torch.zeros(a.total, device=a.device, dtype=torch.float32)
        for i in range(len(a.bucket_names)):
            s, e = a.bucket_range(i)
            tail0 = s + comm_share(e - s, self.world) * self.world
            for lo, hi, off in self._pieces(s, e):
                if lo >= tail0 and self.rank != 0:
                    continue
                full[lo:hi].copy_(compact[off: off + hi - lo])
        if self.world > 1:
            import torch.distributed as dist

            eng = self.engine
            for c0 in range(0, a.total, 1 << 28):
                piece = full[c0: c0 + (1 << 28)]
                if eng.native is not None:
                    eng.native.allreduce_(piece, form="allreduce")
                elif getattr(eng, "staged", False):
                    h = piece.cpu()
                    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=eng.pg)
                    piece.copy_(h.to(piece.device))
                else:
                    dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=eng.pg)
        return full.cpu() if to_cpu else full

    def state_dict(self, to_cpu: bool = True):
        """COLLECTIVE (every rank calls it): the consolidated optimizer state in the replicated layout - {"state": {"master", "m", "v": fp32 [arena.total],
        "t"}, "hyper": {...}} - the same "state" trainer.AfkAdamW.state_dict() saves, so a run may be checkpointed sharded and resumed replicated (or at another
        world size) and vice versa.  One moment at a time (33 GB fp32 for AF3-7B), moved to the host before the next when to_cpu (the default)."""
        self.arena.join_streams()
        return {"state": {"master": self._consolidate(self.master, to_cpu), "m": self._consolidate(self.m, to_cpu), "v": self._consolidate(self.v, to_cpu),
                          "t": self.t},
                "hyper": {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay},
                "layout": {"format": "flat-arena-fp32", "numel": self.arena.total, "saved_from_world": self.world}}

    def load_state_dict(self, sd):
        """restore from the consolidated layout (state_dict() above or trainer.AfkAdamW.state_dict()): every rank slices out the pieces it owns; the bf16
        working copy of EVERY parameter follows the restored fp32 master, as AfkAdamW.load_state_dict does."""
        st = sd["state"]
        a = self.arena
        for name in ("master", "m", "v"):
            if st[name].numel() != a.total:
                raise AfkError(f"ShardedAdamW.load_state_dict: {name} has {st[name].numel()} elements, the arena {a.total}")
        for a0, a1, off in self.owned:
            n = a1 - a0
            self.master[off: off + n].copy_(st["master"][a0:a1])
            self.m[off: off + n].copy_(st["m"][a0:a1])
            self.v[off: off + n].copy_(st["v"][a0:a1])
        self.t = int(st["t"])
        for k, v in (sd.get("hyper") or {}).items():
            setattr(self, k, tuple(v) if k == "betas" else v)
        for c0 in range(0, a.total, 1 << 28):
            a.params.data[c0: c0 + (1 << 28)].copy_(st["master"][c0: c0 + (1 << 28)])
        a.step_counter += 1
        self._mark_synced()
        if a.device.type == "cuda":
            a.refresh_shadows(force=True)

