"""HIP streams of the training step, with queue priorities.

The step runs on three streams: the critical path (forward, dgrad, attention, norms: "compute"), the weight-gradient GEMMs
("wgrad") and the optimizer / gradient exchange ("side").  The compute stream IS the step (392 of 406 ms busy, profiles/r05_bench_timeline_default_schedule.md),
and beside the one-per-CU 256 x 256 wgrad workgroups its kernels ran 2.5-7 x slower than alone.  With priorities the workgroup
dispatcher hands a CU that frees up to the compute queue first; the wgrad GEMMs take what is left.  torch.cuda.Stream only reaches "normal" and
"high", so the handles come from libafk (afk_stream_create: the device's whole range) and are wrapped with torch.cuda.ExternalStream.

AFK_STREAM_PRIORITIES:  "hl" (default) compute at the highest, wgrad / side at the lowest priority;  "h0" compute highest, the others at the default;
"0l" compute at the default, the others lowest;  "0" off (every role a plain torch stream at the default priority: the round-5 schedule).

AFK_SIDE_CUS=n (0 / unset = off): the "side" stream is created with a CU mask of n compute units, n / 8 on every XCD (afk_stream_create_cu_mask), at the default
queue priority.  The optimizer's waves then stop landing on all 256 CUs; the step is power-limited, so the GEMM queues do not miss the CUs (profiles/r04_cu_contention.json).
AFK_MAIN_CUS=m: the "compute" and "wgrad" streams masked to the LAST m CUs (probe: a disjoint partition when n + m <= the CU count; masked streams lose their priority).
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib

_MODES = {"hl": {"compute": "greatest", "wgrad": "least", "side": "least"}, "h0": {"compute": "greatest", "wgrad": "default", "side": "default"},
          "0l": {"compute": "default", "wgrad": "least", "side": "least"}}
_keep = []   # ExternalStream does not own its handle: the handles live as long as the process


def mode() -> str:
    m = os.environ.get("AFK_STREAM_PRIORITIES", "hl")
    m = "hl" if m == "1" else m
    if m != "0" and m not in _MODES:
        raise _lib.AfkError(f"AFK_STREAM_PRIORITIES={m!r}: one of 0, 1, {', '.join(_MODES)}")
    return m


def enabled() -> bool:
    return mode() != "0"


def priority_range():
    """(least, greatest) of the current device: numerically greater = lower priority; (0, 0) = the device has one level"""
    least, greatest = ctypes.c_int(0), ctypes.c_int(0)
    _lib.call("afk_stream_priority_range", ctypes.byref(least), ctypes.byref(greatest))
    return least.value, greatest.value


def make_stream(device, role: str) -> torch.cuda.Stream:
    """a stream for `role` ("compute" | "wgrad" | "side") on `device`; plain torch.cuda.Stream when priorities are off"""
    device = torch.device(device)
    if role not in _MODES["hl"]:
        raise _lib.AfkError(f"make_stream: unknown role {role!r}")
    n_mask = int(os.environ.get("AFK_SIDE_CUS", "0") or 0) if role == "side" else int(os.environ.get("AFK_MAIN_CUS", "0") or 0)
    if n_mask > 0:
        with torch.cuda.device(device):
            total = torch.cuda.get_device_properties(device).multi_processor_count
            n_mask = min(n_mask, total)
            first = 0 if role == "side" else total - n_mask
            h = ctypes.c_void_p(0)
            _lib.call("afk_stream_create_cu_mask", first, n_mask, ctypes.byref(h))
        s = torch.cuda.ExternalStream(h.value, device=device)
        s.afk_role, s.afk_priority, s.afk_cus = role, 0, (first, n_mask)
        _keep.append(h.value)
        return s
    if not enabled():
        return torch.cuda.Stream(device=device)
    with torch.cuda.device(device):
        least, greatest = priority_range()
        want = _MODES[mode()][role]
        prio = least if want == "least" else greatest if want == "greatest" else 0
        h = ctypes.c_void_p(0)
        _lib.call("afk_stream_create", prio, ctypes.byref(h))
    s = torch.cuda.ExternalStream(h.value, device=device)
    s.afk_role, s.afk_priority = role, prio
    _keep.append(h.value)
    return s
