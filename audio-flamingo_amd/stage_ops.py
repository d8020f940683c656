"""The STAGES of the training step as registered PyTorch operators (``torch.ops.afk.<stage>`` / ``torch.ops.afk.<stage>_bwd``).

BASELINE.json north_star: "Python host code on PyTorch-ROCm registers custom ops that call hand-written HIP kernels through a thin C-ABI layer".
Rounds 1-3 ran the step on ``torch.autograd.Function`` stages (functional.py) and registered only an op-level side surface (custom_ops.py).  Since
round 4 the product path IS the registered-op path: every stage ``modeling.forward`` executes - conv stem, encoder layer, pool + LayerNorm,
projector, embedding scatter, decoder layer, final RMSNorm, lm_head, lm_head + loss, Music Flamingo's rotary time embedding - is an operator of the
``afk`` library with
    * a schema (tensor inputs, the parameter arena, the GRADIENT arena declared as MUTATED - weight gradients are written in place, which is what
      ``mutates_args`` is for - and a string key that names the stage's static arguments),
    * a backward operator and an autograd registration (``torch.library.register_autograd``) that stitches the two,
    * a fake (meta) implementation, so that ``torch.compile(fullgraph=True)`` traces THROUGH a training step built from these ops
      (tests/test_custom_ops_gpu.py::test_whole_training_step_traces_under_torch_compile).
The bodies of the stages stay in functional.py (``<Stage>Fn.forward / .backward``: sequences of ``ops.*`` calls into libafk.so); this module is the
dispatcher plumbing around them.  ``<Stage>Fn.apply(...)`` - what modeling.py calls - routes through ``torch.ops.afk.<stage>``.

Static (non-tensor) arguments - the arena, parameter-key prefixes, head counts, eps - travel as ONE string: a deterministic key into a side table
(the key contains every static argument and every tensor shape, so equal keys mean equal stage geometry).  What the backward needs beyond tensors
(``ctx.meta`` of the stage) is a pure function of that key and is remembered per key; activations a stage keeps for backward are OUTPUTS of the
forward operator (a tensor that is an input, or a view of the arenas, is passed by reference instead - operators must not return aliases).
"""
from __future__ import annotations

import weakref
from typing import List, Optional, Sequence

import torch
from torch import Tensor

_LIB = torch.library.Library("afk", "FRAGMENT")
_ARENAS: "weakref.WeakValueDictionary[int, object]" = weakref.WeakValueDictionary()
from collections import OrderedDict

# Both tables are bounded LRU maps: a key contains every tensor shape of its stage, so a data loader with varying batch geometry mints new
# keys for every layer; a key is touched by its forward AND its backward, so the least recently used ones belong to steps long finished.
_TABLE_CAP = 16384
_STATIC = OrderedDict()   # key -> (arena id, static args in forward order)
_CACHE = OrderedDict()    # key -> {"fwd_meta", "src", "out_spec", "bwd_spec"}


_PINNED = set()           # keys a tracer has seen (_fwd_fake / _bwd_fake): a compiled or captured step bakes the key string in as a constant and calls the
                          # operator implementations directly, never through _Stage.apply() - nothing would re-insert the key after an eviction


def _touch(table, key):
    table.move_to_end(key)
    if len(table) > _TABLE_CAP:
        for k in list(table):
            if len(table) <= _TABLE_CAP:
                break
            if k not in _PINNED:
                del table[k]


def _cache_of(key: str):
    c = _CACHE.get(key)
    if c is None:
        raise RuntimeError(f"afk stage {key[:80]!r}...: what its forward kept for backward was evicted (more than {_TABLE_CAP} distinct stage geometries "
                           f"between this stage's forward and its backward) or the forward never ran in this process")
    return c

_STAGES = {}   # stage name -> _Stage
_UID = [0]
_ARENA = object()   # stands for "the arena" inside remembered metadata (the side tables must not keep arenas - 16 GB of HBM each - alive)


def _uid(arena) -> int:
    u = getattr(arena, "_stage_uid", None)
    if u is None:
        _UID[0] += 1
        u = arena._stage_uid = _UID[0]
        _ARENAS[u] = arena
        weakref.finalize(arena, _purge, u)
    return u


def _purge(uid: int):
    tag = f"|{uid}|"
    for table in (_STATIC, _CACHE):
        for k in [k for k in table if tag in k[: k.index("|", k.index("|") + 1) + 1]]:
            table.pop(k, None)
            _PINNED.discard(k)


class _Ctx:
    """what a stage body sees instead of autograd's FunctionCtx"""

    saved_tensors = ()
    meta = None
    needs_input_grad = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


def _same_meta(a, b) -> bool:
    if isinstance(a, tuple) and isinstance(b, tuple):
        return len(a) == len(b) and all((x is y) or (not isinstance(x, Tensor) and not isinstance(y, Tensor) and x == y) for x, y in zip(a, b))
    return a is b or (not isinstance(a, Tensor) and a == b)


def _placeholder(like: Tensor) -> Tensor:
    return like.new_empty(0)


def _arena_of(key: str):
    if key not in _STATIC:
        raise RuntimeError(f"afk stage {key[:80]!r}...: its metadata was evicted (more than {_TABLE_CAP} distinct stage geometries between this stage's forward "
                           f"and its backward)")
    _touch(_STATIC, key)
    aid, static = _STATIC[key]
    arena = _ARENAS.get(aid)
    if arena is None:
        raise RuntimeError(f"afk stage {key!r}: its arena no longer exists")
    return arena, static


class _Stage:
    def __init__(self, name: str, fn, slots: Sequence[str]):
        """slots: one entry per argument of fn.forward (after ctx): "T" tensor, "T?" optional tensor, "A" the arena, "S" static"""
        self.name, self.fn, self.slots = name, fn, tuple(slots)
        self.t_idx = [i for i, s in enumerate(slots) if s in ("T", "T?")]
        self.a_idx = slots.index("A")
        self.s_idx = [i for i, s in enumerate(slots) if s == "S"]
        targs = ", ".join(f"Tensor{'?' if slots[i] == 'T?' else ''} t{k}" for k, i in enumerate(self.t_idx))
        # the forward operator is FUNCTIONAL in its schema (torch.library refuses an autograd formula on an operator that declares mutation); the
        # backward operator - where every weight gradient is written - declares the gradient arena as mutated.  (One forward stage does write
        # into the gradient arena: lm_head + loss produces dW chunk by chunk while the logits chunk exists and parks it there; no other operator
        # reads that slice before the stage's own backward operator, which rescales it.)
        # (the gradient arena rides along as a plain input so that the autograd formula can hand it to the backward operator as a traced tensor)
        _LIB.define(f"{name}({targs}, Tensor params, Tensor grads, str key) -> Tensor[]")
        # (a Tensor[] result beside a mutated argument is refused by torch's functionalisation pass; the input gradients have a fixed count per stage)
        rets = ", ".join(["Tensor"] * len(self.t_idx))
        _LIB.define(f"{name}_bwd(Tensor grad, Tensor[] saved, Tensor params, Tensor(a!) grads, str key) -> ({rets})")
        _LIB.impl(name, self._fwd_impl, "CompositeExplicitAutograd")
        _LIB.impl(f"{name}_bwd", self._bwd_impl, "CompositeExplicitAutograd")
        torch.library.register_fake(f"afk::{name}", self._fwd_fake, lib=_LIB)
        torch.library.register_fake(f"afk::{name}_bwd", self._bwd_fake, lib=_LIB)
        torch.library.register_autograd(f"afk::{name}", self._autograd_backward, setup_context=self._setup_context, lib=_LIB)
        self.op = getattr(torch.ops.afk, name)
        self.op_bwd = getattr(torch.ops.afk, f"{name}_bwd")

    # ------------------------------------------------------------------ call site (what <Stage>Fn.apply does)
    def apply(self, *args):
        args = list(args) + [None] * (len(self.slots) - len(args))   # trailing optional arguments of the stage
        arena = args[self.a_idx]
        tens = [args[i] for i in self.t_idx]
        static = tuple(args[i] for i in self.s_idx)
        shapes = tuple(None if t is None else (tuple(t.shape), str(t.dtype)) for t in tens)
        uid = _uid(arena)
        # does this call record a backward?  (the operator's implementation runs below the autograd key with grad mode off and cannot see it;
        # lm_head + loss produces its gradients during the forward sweep only when somebody will ask for them)
        need = int(torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tens))
        # run-time state that decides WHAT a stage keeps for backward belongs to the key (equal keys must mean equal saved-tensor layout,
        # whatever the interleaving of forwards and backwards): a stage declares it with a static `key_state(arena, *static_args) -> str`
        ks = getattr(self.fn, "key_state", None)
        state = f"|{ks(arena, *static)}" if ks is not None else ""
        key = f"{self.name}|{uid}|{static!r}|{shapes!r}{state}|g{need}"
        if key not in _STATIC:
            _STATIC[key] = (uid, static)
        _touch(_STATIC, key)
        return self.op(*tens, arena.params, arena.grads, key)[0]

    def _full_args(self, tens, arena, static):
        full = [None] * len(self.slots)
        for k, i in enumerate(self.t_idx):
            full[i] = tens[k]
        for k, i in enumerate(self.s_idx):
            full[i] = static[k]
        full[self.a_idx] = arena
        return full

    # ------------------------------------------------------------------ forward operator
    def _fwd_impl(self, *a):
        *tens, params, grads, key = a
        arena, static = _arena_of(key)
        ctx = _Ctx()
        full = self._full_args(tens, arena, static)
        ctx.needs_input_grad = (key.endswith("|g1"),) * len(full)
        out = self.fn.forward(ctx, *full)
        g_store, p_store = grads.untyped_storage().data_ptr(), params.untyped_storage().data_ptr()
        outs, src = [out], []
        for t in ctx.saved_tensors:
            if t is None:
                src.append(("none", 0))
                continue
            k = next((k for k, x in enumerate(tens) if x is t), None)
            if k is not None:
                src.append(("in", k))       # an input kept for backward: passed by reference from the autograd context
            elif t.untyped_storage().data_ptr() in (g_store, p_store) or t is out:
                raise RuntimeError(f"afk::{self.name}: a stage must not keep a view of the arenas (or its own output) for backward")
            else:
                src.append(("out", len(outs)))
                outs.append(t)
        c = _CACHE.setdefault(key, {})
        _touch(_CACHE, key)
        meta = ctx.meta
        fwd_meta = tuple(_ARENA if m is arena else m for m in meta) if isinstance(meta, tuple) else meta
        if "src" in c and (c["src"] != src or not _same_meta(c["fwd_meta"], fwd_meta)):
            # the layout is remembered per key and read by whichever backward of that key runs next: a second layout under one key would hand a
            # backward the wrong saved tensors.  A stage whose layout depends on run-time state must expose that state through key_state()
            raise RuntimeError(f"afk::{self.name}: two forwards with the same stage key kept different things for backward ({c['src']} vs {src}); "
                               f"give the stage a key_state()")
        c["fwd_meta"] = fwd_meta
        c["src"] = src
        c["out_spec"] = [(tuple(t.shape), t.dtype) for t in outs]
        return outs

    def _fwd_fake(self, *a):
        *tens, params, grads, key = a
        c = _CACHE.get(key)
        if c is None:
            raise RuntimeError(f"afk::{self.name}: run the step eagerly once before tracing it (output shapes are recorded per stage geometry)")
        _PINNED.add(key)   # traced: the compiled step will call _fwd_impl / _bwd_impl with this very string for as long as it lives
        return [params.new_empty(shape, dtype=dtype) for shape, dtype in c["out_spec"]]

    # ------------------------------------------------------------------ autograd registration
    def _setup_context(self, ctx, inputs, output):
        *tens, params, grads, key = inputs
        ctx.key = key
        ctx.in_present = [t is not None for t in tens]
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(*[t for t in tens if t is not None], *output[1:])
        # the arenas are carried as plain attributes: they are MEANT to change between this stage's forward and its backward (earlier backward
        # operators write their gradients, declared as mutation), which autograd's version check on saved tensors would refuse
        ctx.arenas = (params, grads)

    def _autograd_backward(self, ctx, gouts):
        key = ctx.key
        n_in = len(ctx.in_present)
        g = gouts[0] if gouts is not None else None
        none = (None,) * (n_in + 3)
        if g is None:
            return none
        sv = list(ctx.saved_tensors)
        params, grads = ctx.arenas
        it = iter(sv)
        tens = [next(it) if p else None for p in ctx.in_present]
        outs = [None] + list(it)                       # outs[i] for i >= 1: the activations the forward operator returned
        saved = []
        for kind, i in _cache_of(key)["src"]:
            t = tens[i] if kind == "in" else outs[i] if kind == "out" else None
            saved.append(_placeholder(g) if t is None else t)
        res = self.op_bwd(g.contiguous(), saved, params, grads, key)
        in_grads = [None if (r.numel() == 0 and not p) or r.numel() == 0 else r for r, p in zip(res, ctx.in_present)]
        return (*in_grads, None, None, None)

    # ------------------------------------------------------------------ backward operator
    def _bwd_impl(self, g, saved, params, grads, key):
        arena, static = _arena_of(key)
        c = _cache_of(key)
        _touch(_CACHE, key)
        ctx = _Ctx()
        ctx.saved_tensors = tuple(None if kind == "none" else t for (kind, _), t in zip(c["src"], saved))
        meta = c["fwd_meta"]
        ctx.meta = tuple(arena if m is _ARENA else m for m in meta) if isinstance(meta, tuple) else meta
        ctx.needs_input_grad = (True,) * len(self.slots)
        res = self.fn.backward(ctx, g)
        if not isinstance(res, tuple):
            res = (res,)
        if grads.data_ptr() != arena.grads.data_ptr():
            # a functionalising tracer (torch.compile without re-inplacing) hands the operator a COPY of the mutated argument and copies it back
            # afterwards; the stage body wrote through the arena's own views, so mirror the arena into that copy (eager calls never get here)
            grads.copy_(arena.grads)
        out = tuple(_placeholder(g) if res[i] is None else res[i] for i in self.t_idx)
        c["bwd_spec"] = [(tuple(t.shape), t.dtype) for t in out]
        return out

    def _bwd_fake(self, g, saved, params, grads, key):
        c = _CACHE.get(key, {})
        if "bwd_spec" not in c:
            raise RuntimeError(f"afk::{self.name}_bwd: run one eager forward + backward before tracing the step")
        _PINNED.add(key)
        return tuple(g.new_empty(shape, dtype=dtype) for shape, dtype in c["bwd_spec"])


def register_stage(name: str, fn, slots: Sequence[str]):
    """make ``fn`` (a class with static ``forward(ctx, ...)`` / ``backward(ctx, grad)`` written against ops.*) the registered operator afk::<name>;
    returns the callable that ``fn.apply`` becomes"""
    st = _Stage(name, fn, slots)
    _STAGES[name] = st
    return st.apply


def registered_stages() -> List[str]:
    return sorted(_STAGES)
