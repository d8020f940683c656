"""Training-loop compatibility layer: run existing fine-tune scripts on the MI355X path.

  * ``AfkAdamW``     - the fused arena optimizer (arena.FusedAdamW: bf16 params + fp32 master / m / v, one launch per decay class)
                       behind the ``torch.optim.Optimizer`` interface: ``param_groups`` (LR schedulers write ``lr`` there),
                       ``step()``, ``zero_grad()``, ``state_dict()`` / ``load_state_dict()`` for checkpoint / resume.
  * ``AfkTrainer``   - ``transformers.Trainer`` subclass: builds ``AfkAdamW`` from the TrainingArguments, never wraps the model in
                       torch DDP (weight gradients are written straight into the gradient arena, so DDP's autograd hooks would never
                       fire) and instead drives ``dp.DataParallelEngine`` (per-layer bucket all-reduce on a side stream, overlapped
                       with backward - the oracle's path is DistributedDataParallel, TORCH/nn/parallel/distributed.py:662-666,
                       reached from TF/trainer.py:1892-1963); honours gradient accumulation (``no_sync`` on non-final micro-steps).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import ops
from .arena import FusedAdamW


class AfkAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.arena = model.arena
        params = [p for p in model.parameters() if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.fused = FusedAdamW(self.arena, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.grad_scale = 1.0  # set to 1/world by a caller that hands over SUMMED data-parallel gradients
        self.gates = None      # data parallel: DataParallelEngine.bucket_gate of the exchange that preceded this step

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        self.fused.lr, self.fused.betas, self.fused.eps = float(g["lr"]), tuple(g["betas"]), float(g["eps"])
        self.fused.step(grad_scale=self.grad_scale, gates=self.gates)
        self.gates = None
        return loss

    def zero_grad(self, set_to_none: bool = True):
        self.arena.zero_grad()

    def state_dict(self):
        f = self.fused
        return {"state": {"master": f.master, "m": f.m, "v": f.v, "t": f.t},
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        f = self.fused
        f.master.copy_(sd["state"]["master"]), f.m.copy_(sd["state"]["m"]), f.v.copy_(sd["state"]["v"])
        f.t = int(sd["state"]["t"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
        self.arena.params.copy_(f.master)  # bf16 working copy follows the fp32 master
        self.arena.step_counter += 1
        f._mark_synced()  # the fp32 master just restored is the truth; do not re-derive it from the bf16 working copy
        self.arena.refresh_shadows(force=True)


def unwrap_optimizer(opt):
    """peel accelerate's AcceleratedOptimizer (and any other `.optimizer`-holding wrapper) off: the HF Trainer loop never hands out the
    optimizer it was given, it hands out the wrapper (the DP gates / master sync must reach the AfkAdamW underneath)"""
    seen = 0
    while opt is not None and not isinstance(opt, AfkAdamW) and hasattr(opt, "optimizer") and seen < 8:
        opt, seen = opt.optimizer, seen + 1
    return opt if isinstance(opt, AfkAdamW) else None


def _trainer_base():
    from transformers import Trainer

    return Trainer


class AfkTrainer(_trainer_base()):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        # the gradient exchange is ours (DataParallelEngine): accelerate must not wrap the model in DistributedDataParallel
        self.accelerator.prepare_model = lambda model, device_placement=None, evaluation_mode=False: model
        self._afk_engine = None
        self._afk_scale = None
        # max_grad_norm (HF default 1.0): the reference clips with torch.nn.utils.clip_grad_norm_ - a foreach norm over ~700 gradient tensors
        # plus a read-modify-write pass.  Here the request is only RECORDED: the fused optimizer step takes the norm in one streaming pass over
        # the flat gradient arena and applies the coefficient inside its AdamW launches (arena.FusedAdamW.clip_norm).  The returned device
        # scalar is filled by that step, i.e. before the Trainer reads it for logging.
        stock_clip = self.accelerator.clip_grad_norm_

        def _clip(parameters, max_norm, norm_type=2):
            opt = unwrap_optimizer(self.optimizer)
            if opt is not None and opt.fused.hyper is not None and norm_type == 2:
                opt.fused.clip_norm = float(max_norm)
                return opt.fused.grad_norm
            return stock_clip(parameters, max_norm, norm_type)

        self.accelerator.clip_grad_norm_ = _clip

    def create_optimizer(self, model=None):
        if self.optimizer is None:
            a = self.args
            self.optimizer = AfkAdamW(self.model, lr=a.learning_rate, betas=(a.adam_beta1, a.adam_beta2), eps=a.adam_epsilon,
                                      weight_decay=a.weight_decay)
        return self.optimizer

    def _engine(self):
        if self._afk_engine is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from .dp import DataParallelEngine

            self._afk_engine = DataParallelEngine(self.model.arena, overlap=True)
            if self._afk_engine.sharded:
                # AFK_DP_FORM=rs_adamw_ag reduce-SCATTERS the gradients: only arena.ShardedAdamW (engine.make_optimizer) may consume them.  The Trainer's
                # optimizer is the replicated AfkAdamW (its state_dict is the full fp32 state the HF checkpoint format expects) - refuse instead of
                # training on gradient shares that were never reduced
                from ._lib import AfkError

                raise AfkError("AfkTrainer drives the replicated optimizer (AfkAdamW); AFK_DP_FORM=rs_adamw_ag (optimizer sharded over the ranks) is "
                               "available through DataParallelEngine.make_optimizer() / bench.py, not through the Trainer - unset it")
            self._afk_engine.broadcast_parameters(0)
            opt = unwrap_optimizer(self.optimizer)
            if opt is not None:
                opt.fused.sync_master()
            self._afk_scale = torch.full((1,), 1.0 / self._afk_engine.world, device=self.model.arena.device, dtype=torch.float32)
        return self._afk_engine

    def training_step(self, model, inputs, num_items_in_batch=None):
        eng = self._engine()
        if eng is None:
            return super().training_step(model, inputs, num_items_in_batch)
        sync = bool(getattr(self.accelerator, "sync_gradients", True))
        eng.enabled = sync
        eng.begin_backward()
        try:
            loss = super().training_step(model, inputs, num_items_in_batch)
            if sync:
                eng.finish()
                g = self.model.arena.grads
                ops.scale_add_(g, g, self._afk_scale, accumulate=False)  # averaged gradients: clipping / logging see the DDP convention
                opt = unwrap_optimizer(self.optimizer)
                if opt is not None:
                    # a bucket NO rank touched this step (e.g. the audio tower on an all-text step) must keep parameters AND moments, as
                    # torch.optim does for `grad is None`: the AdamW launches of the coming optimizer.step() are gated on these flags
                    opt.gates = eng.bucket_gate
        finally:
            eng.enabled = True
        return loss
