"""The optimiser step of main.train (main.py:263-266: `clip_grad_norm_(model.parameters(), opt.grad_clip)` followed by
`optimizer.step()` of torch.optim.Adam) on the library's own multi-tensor kernels (csrc/optim.hip).

`ClipAdam` IS a torch.optim.Adam (same constructor, param groups, `state` entries 'step' / 'exp_avg' / 'exp_avg_sq',
`state_dict()` / `load_state_dict()` interchangeable with the reference's optimizer); only `step()` is replaced, and
`step_clipped(max_norm)` fuses the gradient clipping into it: the total norm comes from ordered per-chunk partial sums,
the clip factor stays on the device, and every gradient is scaled while the Adam pass reads it - the clipped gradients are
never written back (so, unlike after clip_grad_norm_, `.grad` still holds the UNCLIPPED gradients afterwards).
"""
import ctypes as C
import math

import torch

from .hip import OPT_CHUNK, OPT_MAX_TENSORS, GvdHipError, OptGroup, check, lib, ptr, stream_ptr


class ClipAdam(torch.optim.Adam):
    def __init__(self, params, **kw):
        for k in ('amsgrad', 'maximize', 'capturable', 'differentiable', 'decoupled_weight_decay'):
            if kw.get(k):
                raise ValueError('ClipAdam: %s is not supported' % k)
        kw.pop('fused', None)
        kw.pop('foreach', None)
        super().__init__(params, foreach=False, fused=False, **kw)
        self._partials = None
        self._clip = None
        self._stepped = []          # state dicts whose step counter the last step_clipped() advanced (rollback)

    # ------------------------------------------------------------------------------------------------------------
    def _live(self):
        """[(param, grad, group)] of the parameters that take part in this step (those with a gradient), in group order."""
        out = []
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or p.dtype != torch.float32 or not p.is_cuda:
                    raise GvdHipError('ClipAdam: dense fp32 GPU parameters only (there is no CPU path)')
                if not p.is_contiguous():
                    raise GvdHipError('ClipAdam: parameters must be contiguous')
                if not p.grad.is_contiguous():       # (autograd lays gradients out like their parameter; be safe)
                    p.grad = p.grad.contiguous()
                out.append((p, p.grad, group))
        return out

    @staticmethod
    def _launches(items):
        """Split [(tensors..., n)] into launches of at most OPT_MAX_TENSORS tensors; yields (first_chunk, OptGroup, idx)."""
        first = 0
        for i0 in range(0, len(items), OPT_MAX_TENSORS):
            part = items[i0:i0 + OPT_MAX_TENSORS]
            g = OptGroup()
            g.count = len(part)
            g.part0 = first
            c = 0
            for t, it in enumerate(part):
                g.chunk0[t] = c
                c += -(-it['n'] // OPT_CHUNK)
                g.n[t] = it['n']
                g.g[t] = it['g'].data_ptr()
                ok = it['g'].data_ptr() % 16 == 0
                for key in ('p', 'm', 'v'):
                    if key in it:
                        getattr(g, key)[t] = it[key].data_ptr()
                        ok = ok and it[key].data_ptr() % 16 == 0
                g.vec_ok[t] = 1 if ok else 0
                g.lr[t] = it.get('lr', 0.0)
                g.bc1[t] = it.get('bc1', 1.0)
                g.bc2_sqrt[t] = it.get('bc2_sqrt', 1.0)
            g.chunk0[len(part)] = c
            yield g
            first += c

    def grad_norm_and_clip(self, live, max_norm):
        """Device tensor [2] = (total L2 norm of the live gradients, min(1, max_norm / (norm + 1e-6)))."""
        items = [{'g': g, 'n': g.numel()} for _, g, _ in live]
        total = sum(-(-it['n'] // OPT_CHUNK) for it in items)
        dev = live[0][0].device
        if self._partials is None or self._partials.numel() < total or self._partials.device != dev:
            self._partials = torch.empty(total, dtype=torch.float32, device=dev)
        self._clip = torch.empty(2, dtype=torch.float32, device=dev)     # (fresh: a caller may keep last step's norm)
        for g in self._launches(items):
            check(lib().gvd_sumsq_partials(C.byref(g), ptr(self._partials), stream_ptr()), 'gvd_sumsq_partials')
        check(lib().gvd_clip_coef(ptr(self._partials), total, float(max_norm), ptr(self._clip), stream_ptr()),
              'gvd_clip_coef')
        return self._clip

    @torch.no_grad()
    def step_clipped(self, max_norm=None, closure=None, skip=None):
        """clip_grad_norm_(all parameters of all groups, max_norm) + Adam.step() in one go.  Returns the pre-clip total
        gradient norm as a device scalar (None without clipping).
        skip: optional device int32 vector; if any element is non-zero WHEN THE KERNELS RUN, the update kernels change
        nothing (the norm is still computed).  The caller reads the vector afterwards and, if it was raised, calls
        rollback_step_counts() - so the optimiser can be enqueued before the flags that validate the step are known."""
        if closure is not None:
            raise ValueError('ClipAdam: closures are not supported')
        live = self._live()
        self._stepped = []
        if not live:
            return None
        if skip is not None:
            assert skip.is_cuda and skip.dtype == torch.int32 and skip.is_contiguous() and 0 < skip.numel() <= 64
        clip = self.grad_norm_and_clip(live, max_norm) if max_norm is not None else None
        # one launch list per (betas, eps, weight_decay) combination (the recipe has one: main.py:660-677)
        by_hyper = {}
        for p, g, group in live:
            st = self.state[p]
            if len(st) == 0:
                st['step'] = torch.tensor(0.0, dtype=torch.float32)
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if st['step'].is_cuda:       # (a state_dict written by torch's fused Adam keeps `step` on the device)
                st['step'] = st['step'].detach().cpu()
            st['step'] += 1
            self._stepped.append(st)
            t = float(st['step'])
            b1, b2 = group['betas']
            key = (b1, b2, group['eps'], group['weight_decay'])
            by_hyper.setdefault(key, []).append({
                'p': p, 'g': g, 'm': st['exp_avg'], 'v': st['exp_avg_sq'], 'n': p.numel(), 'lr': float(group['lr']),
                'bc1': 1.0 - b1 ** t, 'bc2_sqrt': math.sqrt(1.0 - b2 ** t)})
        for (b1, b2, eps, wd), items in by_hyper.items():
            for g in self._launches(items):
                check(lib().gvd_adam_step(C.byref(g), ptr(clip), ptr(skip), 0 if skip is None else skip.numel(), b1, b2, eps,
                                          wd, stream_ptr()), 'gvd_adam_step')
        # the kernels wrote p / exp_avg / exp_avg_sq through raw pointers: tell autograd's version counters, as torch's
        # in-place Adam does - anything keyed on `_version` (att_model._packed: the re-laid-out inference weights; saved
        # tensors of a retained graph) must see that the parameters changed
        torch.autograd.graph.increment_version([t for p, _, _ in live
                                                for t in (p, self.state[p]['exp_avg'], self.state[p]['exp_avg_sq'])])
        return None if clip is None else clip[0]

    def rollback_step_counts(self):
        """Undo the step-counter advance of the last step_clipped() (whose device side was skipped: `skip` was raised)."""
        for st in self._stepped:
            st['step'] -= 1
        self._stepped = []

    def step(self, closure=None):
        self.step_clipped(None, closure)
