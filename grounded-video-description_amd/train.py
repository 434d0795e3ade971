"""One optimisation step with the reference's recipe (main.train, main.py:234-266, 660-677):
loss = (lm + w_att2*att2 + w_grd*grd + w_cls*cls) / n_replicas, clip_grad_norm_(0.1), Adam(lr 5e-4; x0.1
for the fc7 / vis_embed parameters).  Under torch.distributed the gradients are averaged over ranks
(dist.GradAllReducer) between backward and the clip, which is exactly the reference's DataParallel
semantics."""
import os

import torch
import torch.nn as nn

from . import dist as gdist


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
        on_gpu = all(p.is_cuda for g in groups for p in g['params'])
        if on_gpu and os.environ.get('GVD_OWN_ADAM', '1') == '1':
            # clip + Adam on the library's own multi-tensor kernels (optim.ClipAdam is a torch.optim.Adam: same state,
            # same state_dict); GVD_OWN_ADAM=0: clip_grad_norm_ + torch's fused Adam
            from .optim import ClipAdam
            return ClipAdam(groups)
        return torch.optim.Adam(groups, fused=True) if on_gpu else torch.optim.Adam(groups)
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
    def __init__(self, model, opt, bucket_mb=64):
        self.model, self.opt = model, opt
        self.optimizer = build_optimizer(model, opt)
        self.reducer = gdist.GradAllReducer(model, bucket_mb=bucket_mb)
        self._grad_norm = None

    @property
    def last_grad_norm(self):
        """Pre-clip total gradient norm of the last step (main.py:265).  Kept on the device; read on demand."""
        return None if self._grad_norm is None else float(self._grad_norm)

    def step(self, args):
        """args: the 11 positional tensors of AttModel.forward.  Returns the 4 detached losses [4].

        ONE device->host read per step: the kernel-status counts of the persistent kernels (the bi-GRU runs as one in
        training too: a grid-barrier timeout must not reach the optimizer silently) - MAX-reduced over the ranks together
        with the reducer's rediscovery flag, so that under data parallelism every rank raises (or rebuilds its buckets)
        in the same step instead of leaving its peers blocked in the next collective.  The gradient norm stays a device
        tensor (clip_grad_norm_ scales on the device).  GVD_TRAIN_DEFER_STATUS=1 (own optimiser only): the read happens
        AFTER clip + Adam were enqueued, predicated on the device by the same word (see _finish_step_deferred; measured:
        92.86 vs 92.95 ms per batch_size = 64 step - the host is far enough ahead of the GPU for the read not to matter, so
        the plain order stays the default)."""
        self.model.zero_grad(set_to_none=True)
        self.reducer.reset()
        losses = self.model(*args, 'MLE')
        loss = combine_losses(losses, self.opt)
        loss.backward()
        counts = self.model.kernel_status_counts() if hasattr(self.model, 'kernel_status_counts') else None
        own = hasattr(self.optimizer, 'step_clipped')
        if own and os.environ.get('GVD_TRAIN_DEFER_STATUS', '0') == '1':
            return self._finish_step_deferred(losses, loss, counts, args)
        if self.reducer.active:
            st = torch.zeros(2, dtype=torch.int32, device=loss.device) if counts is None else counts.to(torch.int32)
            word = self.reducer.finish(status=st, defer=True).tolist()          # the step's one host read
            self.reducer.resolve(word[0])
            bad, contract = word[1], word[2]
        else:
            bad, contract = (0, 0) if counts is None else counts.tolist()       # the step's one host read
        if contract and not bad and self._drop_train_compaction():
            return self.step(args)
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

    def _drop_train_compaction(self):
        """A step on the compacted training layout (GVD_TRAIN_COMPACT=1, train_compact.py) met masked proposals that are
        not zero rows - inputs the reference accepts: compute, don't raise - switch this model to the full row set for
        good and tell the caller to run the step again.  (The status word is the same on every rank, so every rank does.)"""
        if os.environ.get('GVD_TRAIN_COMPACT', '0') != '1' or getattr(self.model, '_train_compact_off', False):
            return False
        self.model._train_compact_off = True
        return True

    def _finish_step_deferred(self, losses, loss, counts, args):
        """Tail of a step with the library's own optimiser: clip + Adam are ENQUEUED first, predicated on the device by the
        step's status word (kernel-status counts of the persistent kernels; under data parallelism the reducer's
        MAX-reduced word: rediscovery flag + those counts), and only then does the host read that word - the step's one
        host read no longer drains the queue in front of the optimiser, whose ~10 launches (and the host work of packing
        them) used to run with the GPU idle.  A raised word means the device skipped the update: the step counters are
        rolled back, and the host either raises (kernel error: same step on every rank) or - rediscovery of a late-used
        parameter - lets the reducer rebuild its buckets and runs the optimiser again, unpredicated."""
        if self.reducer.active:
            st = torch.zeros(2, dtype=torch.int32, device=loss.device) if counts is None else counts.to(torch.int32)
            word = self.reducer.finish(status=st, defer=True)                    # device int32 [3], same on every rank
        else:
            word = None if counts is None else counts.to(torch.int32).contiguous()
        self._grad_norm = self.optimizer.step_clipped(self.opt.grad_clip, skip=word)
        if word is not None:
            host = word.tolist()                                                 # the step's one host read
            flag, bad, contract = (host[0], host[1], host[2]) if self.reducer.active else (0, host[0], host[1])
            if flag or bad or contract:
                self.optimizer.rollback_step_counts()
                if self.reducer.active:
                    self.reducer.resolve(flag)
                if contract and not bad and self._drop_train_compaction():
                    return self.step(args)
                if bad or contract:
                    self.model.raise_for_status(bad, contract)
                self._grad_norm = self.optimizer.step_clipped(self.opt.grad_clip)
        return torch.cat([l.detach() for l in losses])
