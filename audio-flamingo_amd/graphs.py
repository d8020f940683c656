"""HIP-graph capture of one whole training step (frontend -> forward -> backward -> per-bucket AdamW + W^T shadow refresh).

Why: a step of AF3-7B enqueues ~3 000 kernel launches from Python (ctypes -> libafk.so) on three HIP streams; on an idle GPU the host
needs ~41 ms per step for that (bench.py `host_enqueue_ms_idle_gpu`; the 220-414 ms seen inside a timed loop are queue back-pressure),
9 % of the GPU's 440 ms - a limit that would bind as the kernels get faster or the batch smaller.  Shapes are static in training, so the
launch sequence of a step - including the fork / join of the weight-gradient stream and of the optimizer side stream, which become graph
dependencies - is captured ONCE and replayed with a single hipGraphLaunch (9 ms of host time per step, measured).  Same kernels, same order per stream, same arithmetic: the replayed step is
bit-identical to the eager step (tests/test_model_gpu.py::test_graphed_step_matches_eager).

What changes between replays lives in device memory, never in kernel arguments:
    * the inputs (waveforms / ids / labels): static tensors, refill them in place with ``copy_`` before ``replay()``;
    * the optimizer's step-dependent scalars (learning rate, Adam bias corrections): FusedAdamW.hyper, refreshed by
      ``FusedAdamW.advance()`` right before each replay.
Not capturable (use the eager step): data-dependent host decisions - ``check_placeholders`` (host sync), labels / masks whose
valid-row count changes from step to step, gradient checkpointing via torch.utils.checkpoint, and (not validated on this pool's
1-GPU boxes) RCCL collectives inside the capture - bench.py keeps the eager path for world > 1.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from ._lib import AfkError


class GraphedTrainStep:
    def __init__(self, model, optimizer, overlap, step_body: Callable[[], torch.Tensor], warmup: int = 2, stream: Optional[torch.cuda.Stream] = None):
        """step_body(): one EAGER training step on static input tensors, built from
               model.arena.zero_grad(); overlap.begin_step(); loss = model(...).loss; loss.backward(); overlap.finish(); return loss
        It runs `warmup` times eagerly (real optimizer steps: lazy one-time initialisation inside the library and the allocator's
        steady state happen outside the capture), then once more under capture.  `stream`: the stream the capture (and the eager warm-up) runs on -
        pass streams.make_stream(device, "compute") so that the captured critical path keeps its queue priority over the wgrad / optimizer branches."""
        if model.device_.type != "cuda":
            raise AfkError("GraphedTrainStep needs a HIP device")
        if getattr(model, "check_placeholders", False):
            raise AfkError("GraphedTrainStep: set model.check_placeholders = False (its count check is a host sync)")
        if model.gradient_checkpointing:
            raise AfkError("GraphedTrainStep: activation checkpointing (torch.utils.checkpoint) is not captured; use the eager step")
        self.model, self.opt, self.overlap = model, optimizer, overlap
        cur = torch.cuda.current_stream()
        s = stream if stream is not None else torch.cuda.Stream(device=model.device_)
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            for _ in range(max(warmup, 1)):
                step_body()
        cur.wait_stream(s)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()  # hand the eager steps' cached activation blocks back: the graph brings its own pool
        self.graph = torch.cuda.CUDAGraph()
        optimizer._in_capture = True   # the capture pass must not advance t nor record the write of the step scalars into the graph
        try:
            # a live process group has a watchdog THREAD that polls its work events; under the default "global" capture mode such a call from another thread
            # while this one captures invalidates the capture and the watchdog's exception aborts the process (seen 2 / 8 runs of the 1-rank RCCL graph test)
            import torch.distributed as dist
            mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
            with torch.cuda.graph(self.graph, capture_error_mode=mode, **({"stream": stream} if stream is not None else {})):
                self.loss = step_body()
        finally:
            optimizer._in_capture = False
        self.steps = 0

    def __call__(self) -> torch.Tensor:
        self.opt._check_master()
        self.opt.advance()          # t += 1, (lr, bias corrections) -> device, on the stream the graph launches on
        self.graph.replay()
        # host-side bookkeeping the eager path does in finish(): W^T shadows were refreshed inside the graph
        self.model.arena.step_counter += 1
        self.opt._mark_synced()
        for b in self.model.arena.order:
            if b.shadow_kind is not None and not b.shadow_lazy and not (self.model.arena.lazy_T_shadows and b.shadow_kind == "T"):
                b.shadow_version = self.model.arena._version_of(b)
        self.steps += 1
        return self.loss
