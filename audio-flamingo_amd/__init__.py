"""audio_flamingo_amd - MI355X-native (gfx950) training hot path of Audio Flamingo 3.

Layout:
  csrc/        hand-written HIP kernels + the C ABI (include/afk.h) -> lib/libafk.so
  _lib.py      ctypes binding of the C ABI (fails loudly when the library is missing)
  ops.py       tensor-level wrappers (device pointers, strides, current HIP stream)
  functional.py autograd Functions with hand-written backward passes
  modeling.py  nn.Modules mirroring transformers' AudioFlamingo3ForConditionalGeneration surface
  arena.py     flat bf16 parameter/gradient arenas, fp32 AdamW state, W^T shadows
  dp.py        data-parallel gradient all-reduce (RCCL via torch.distributed), bucketed + overlapped
  frontend.py  log-mel frontend tables + wrapper
"""
__version__ = "0.1.0"

# multi-stream schedule (arena.enable_wgrad_stream, dp.BackwardOverlap) + RCCL's stream: more hardware queues than ROCm's default 4 keep
# them from sharing a queue; only effective if the HIP runtime has not initialised yet (see bench.py)
import os as _os

_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
