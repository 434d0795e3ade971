"""One optimisation step with the reference's recipe (main.train, main.py:234-266, 660-677):
loss = (lm + w_att2*att2 + w_grd*grd + w_cls*cls) / n_replicas, clip_grad_norm_(0.1), Adam(lr 5e-4; x0.1
for the fc7 / vis_embed parameters).  Under torch.distributed the gradients are averaged over ranks
(dist.GradAllReducer) between backward and the clip, which is exactly the reference's DataParallel
semantics."""
import os

import torch
import torch.nn as nn

from . import dist as gdist
from . import ops


def build_optimizer(model, opt):
    """main.py:660-677: learning rate x0.1 for the parameters whose name contains 'ctx2pool_grd' or 'vis_embed' (the
    Detectron-transferred tensors), the plain rate for the rest.  The reference makes one param group PER PARAMETER (81
    groups -> ~6 tiny kernels per parameter and step); the update of a parameter depends only on its own group's
    hyper-parameters, so the same parameters are put into TWO groups here and Adam runs as one fused multi-tensor
    kernel per group (identical arithmetic per element; tests/golden/step_*.npz pins one step against the reference)."""
    fine, rest = [], []
    for key, value in dict(model.named_parameters()).items():
        if not value.requires_grad:
            continue
        (fine if ('ctx2pool_grd' in key or 'vis_embed' in key) else rest).append(value)
    groups = [{'params': p, 'lr': opt.learning_rate * s, 'weight_decay': opt.weight_decay,
               'betas': (opt.optim_alpha, opt.optim_beta)} for p, s in ((rest, 1.0), (fine, 0.1)) if p]
    if opt.optim == 'adam':
        if all(p.is_cuda for g in groups for p in g['params']):
            # clip + Adam on the library's own multi-tensor kernels (optim.ClipAdam IS a torch.optim.Adam: same state,
            # same state_dict)
            from .optim import ClipAdam
            return ClipAdam(groups)
        return torch.optim.Adam(groups)      # (CPU parameters: the control-flow tests)
    if opt.optim == 'sgd':
        return torch.optim.SGD(groups, momentum=0.9)
    if opt.optim == 'adamax':
        return torch.optim.Adamax(groups)
    raise ValueError(opt.optim)


def combine_losses(losses, opt):
    """main.py:238-255 for one replica (lm_loss.numel() == 1)."""
    lm, att2, grd, cls = losses
    loss = lm.sum()
    if opt.w_att2:
        loss = loss + opt.w_att2 * att2.sum()
    if opt.w_grd:
        loss = loss + opt.w_grd * grd.sum()
    if opt.w_cls:
        loss = loss + opt.w_cls * cls.sum()
    return loss / lm.numel()


class Trainer:
    def __init__(self, model, opt, bucket_mb=64, compact_rows=None):
        """compact_rows: run the step on the compacted training layout (train_compact.py: per segment its valid proposals
        + one weighted representative of the masked ones) - True / False, or None = the GVD_TRAIN_COMPACT environment
        variable (default off; DESIGN.md section 5 has the evidence and the reason it stays an opt-in)."""
        self.model, self.opt = model, opt
        if compact_rows is not None:
            model.train_compact = bool(compact_rows)
        self.optimizer = build_optimizer(model, opt)
        self.reducer = gdist.GradAllReducer(model, bucket_mb=bucket_mb)
        self._grad_norm = None

    @property
    def last_grad_norm(self):
        """Pre-clip total gradient norm of the last step (main.py:265).  Kept on the device; read on demand."""
        return None if self._grad_norm is None else float(self._grad_norm)

    def step(self, args, _retried=False):
        """args: the 11 positional tensors of AttModel.forward.  Returns the 4 detached losses [4].

        ONE device->host read per step: the kernel-status counts of the persistent kernels (the bi-GRU runs as one in
        training too: a grid-barrier timeout must not reach the optimizer silently) - MAX-reduced over the ranks together
        with the reducer's rediscovery flag, so that under data parallelism every rank raises (or rebuilds its buckets)
        in the same step instead of leaving its peers blocked in the next collective.  The gradient norm stays a device
        tensor (the clip factor is computed and applied on the device)."""
        self.model.zero_grad(set_to_none=True)
        self.reducer.reset()
        if hasattr(self.model, 'kernel_status_counts'):
            # status words left behind by earlier inference calls through the private _sample (a caller that never checked
            # them) are not this step's: they must neither fail it nor be mistaken for a compaction-contract violation
            self.model.kernel_status_counts()
        losses = self.model(*args, 'MLE')
        loss = combine_losses(losses, self.opt)
        loss.backward()
        counts = self.model.kernel_status_counts() if hasattr(self.model, 'kernel_status_counts') else None
        own = hasattr(self.optimizer, 'step_clipped')
        if self.reducer.active:
            st = torch.zeros(2, dtype=torch.int32, device=loss.device) if counts is None else counts.to(torch.int32)
            word = self.reducer.finish(status=st, defer=True).tolist()          # the step's one host read
            self.reducer.resolve(word[0])
            bad, contract = word[1], word[2]
        else:
            bad, contract = (0, 0) if counts is None else counts.tolist()       # the step's one host read
        if self._recompute(bad, contract, _retried):
            return self.step(args, _retried=True)
        if bad or contract:
            self.model.raise_for_status(bad, contract)
        if own:
            # main.py:265-266 in one go: norm from ordered partials, clip factor on the device, gradients scaled while Adam
            # reads them (optim.ClipAdam; every parameter of the model is in one of its groups, as in main.py:660-677)
            self._grad_norm = self.optimizer.step_clipped(self.opt.grad_clip)
        else:
            self._grad_norm = nn.utils.clip_grad_norm_(self.model.parameters(), self.opt.grad_clip)
            self.optimizer.step()
        return torch.cat([l.detach() for l in losses])

    def _recompute(self, bad, contract, retried=False):
        """The two conditions of a step a reference user never sees are COMPUTED, not raised: a grid-barrier timeout of the
        persistent bi-GRU kernel (shared GPU) switches the persistent kernels off for the process, masked proposals that
        are not zero rows switch the compacted training layout off for this model; either way the caller runs the step
        again (forward + backward: fresh dropout draws - the invalid attempt updated nothing).

        The decision uses ONLY rank-identical information - the MAX-reduced status word and whether THIS step was already
        retried - never process-local state such as ops.persistent_kernels_enabled(): a rank that had switched its
        persistent kernels off earlier (say a rank-0-only validation pass that timed out) would otherwise raise while its
        peers retry and block in the next all-reduce.  disable_persistent_kernels is idempotent; a step that is still
        invalid after its one retry raises on every rank."""
        if retried or not (bad or contract):
            return False
        again = False
        if bad:
            ops.disable_persistent_kernels(bad)
            again = True
        if contract and self._drop_train_compaction():
            again = True
        return again

    def _drop_train_compaction(self):
        """A step on the compacted training layout (GVD_TRAIN_COMPACT=1, train_compact.py) met masked proposals that are
        not zero rows - inputs the reference accepts: compute, don't raise - switch this model to the full row set for
        good and tell the caller to run the step again.  (The status word is the same on every rank, so every rank does.)"""
        from .train_step import train_compact_enabled
        if not train_compact_enabled(self.model):
            return False
        self.model._train_compact_off = True
        return True
