"""Batch data parallelism: one process per GPU, RCCL all-reduce of gradients over xGMI.

Replaces the reference's single-process nn.DataParallel (main.py:654-655), which every step broadcasts
all parameters from GPU0, scatters the inputs from GPU0, and reduces all gradients back to GPU0
(SURVEY.md §2.3 C1-C4).  Here replicas are persistent (no weight broadcast after construction), each rank
reads its own shard of segments, and the only collective is a bucketed gradient all-reduce that is
launched from autograd hooks as soon as a bucket's gradients exist, so it overlaps the rest of backward.

Semantics: the reference's DP loss is the mean over replicas of per-replica masked
means (main.py:239-255), so averaging per-rank gradients (sum, then / world) reproduces it exactly.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce is per-link bound, so buckets are
large (default 64 MiB: ~275 MB of fp32 gradients -> 5 collectives) rather than many small ones.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/LOCAL_RANK (torch.distributed.run); no-op for 1 rank."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'   # 'nccl' is RCCL on ROCm
        kw = {}
        if backend == 'nccl':
            torch.cuda.set_device(local)
            kw['device_id'] = torch.device('cuda', local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def broadcast_parameters(module, src=0):
    """One-time sync of the replicas (instead of DataParallel's per-forward broadcast)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


class GradAllReducer:
    """Bucketed, backward-overlapped gradient averaging for a replica.

    Parameters are packed into flat buckets in the order in which their gradients became ready in the first step
    (rank 0's hook order, broadcast: the same layout on every rank) - NOT in reverse registration order, which puts the
    Detectron-transferred fc7 layer (registered last but one, its gradient the very LAST of backward) at the head of
    the second bucket and so holds every later bucket back until backward ends.  A post-accumulate-grad hook marks a
    parameter ready; a bucket whose gradients are all there is copied into its flat buffer and all-reduced
    asynchronously.  Collectives are issued STRICTLY IN
    BUCKET ORDER (a ready bucket waits for its predecessors), so every rank launches the same sequence
    whatever the local hook timing.  `finish()` launches what is left, waits, divides by the world size and
    writes the averages back into `.grad`.

    Parameters that never receive a gradient (the reference's unused core.i2h_2 / core.h2h_2,
    AttModel.py:130-131) are found in the first step — the union over ranks of "has a gradient", one small
    all-reduce(MAX) — and are then left out of the buckets: they no longer hold a bucket's pending count
    above zero (which used to push the bucket with the earliest gradients into `finish()`), and their
    `.grad` stays None exactly as in a single-process run, so the optimizer treats them the same on 1 and
    N ranks.  A parameter that has a gradient on some rank but not on this one contributes zeros.

    The bucket layout is COLLECTIVE state: it only ever changes at a point every rank reaches together.  If an
    excluded parameter receives a gradient later on ANY rank (a rank-asymmetric use), that rank only raises a
    local flag; `finish()` MAX-reduces a small status word on every rank in every step (flag + optional caller
    status, e.g. the kernel-status count of train.Trainer), and when the reduced flag is set ALL ranks - after the
    current buckets were drained, never with collectives in flight - re-run the discovery together, average the
    gradients of the newly used parameters in the same step, and rebuild the buckets for the next one.
    """

    def __init__(self, module, bucket_mb=64, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucket_mb = bucket_mb
        self.all_params = [p for p in module.parameters() if p.requires_grad]
        self.params = list(self.all_params)
        self._discovered = False
        self._fired = []            # step 0: parameters in the order their gradient hooks fired
        self._late = False          # an excluded parameter got a gradient on THIS rank since the last finish()
        self._word = None
        self.trace = False          # record a stream event at reset(), at every bucket launch and at finish() (bench.py)
        self._tev = None
        self.rediscoveries = 0
        self._hooks = []
        self._handles = []
        # GVD_DP_FORCE=1: run the hook/bucket/all-reduce machinery even on a 1-rank group (single-GPU test of the path)
        self.active = self.world > 1 or (os.environ.get('GVD_DP_FORCE') == '1' and dist.is_initialized())
        self._build_buckets()
        if self.active:
            for p in self.all_params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _build_buckets(self):
        if self._handles:
            raise RuntimeError('GradAllReducer: bucket layout changed with %d all-reduce(s) in flight' % len(self._handles))
        self.buckets = []          # list of dict(params, offsets, numel, flat)
        cap = int(self.bucket_mb * 1024 * 1024 / 4)
        cur, cur_n = [], 0
        for p in (self.params if self._discovered else reversed(self.params)):
            if cur and cur_n + p.numel() > cap:
                self._close(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self._close(cur)
        self.where = {}
        for bi, b in enumerate(self.buckets):
            for p in b['params']:
                self.where[id(p)] = bi
        self.reset()

    def _close(self, params):
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += p.numel()
        p0 = params[0]
        self.buckets.append(dict(params=params, offsets=offs, numel=n,
                                 flat=torch.zeros(n, dtype=p0.dtype, device=p0.device)))

    def reset(self):
        """Start of a step (before backward).  Never drops collectives that are still in flight: they are waited for."""
        for _, h in self._handles:
            h.wait()
        self._pending = [len(b['params']) for b in self.buckets]
        self._next = 0             # buckets [0, _next) have been launched
        self._handles = []
        if self.trace and self.all_params and self.all_params[0].is_cuda:
            self._tev = {'t0': self._event(), 'launch': [], 'finish': None, 'reduced': [], 'drained': None}

    @staticmethod
    def _event():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def launch_timeline(self):
        """After a traced step (trace = True) and a device sync: when, on the compute stream's clock, each bucket's
        all-reduce was issued relative to the step's start and to the end of backward -> the window the collective has to
        hide in.  [{'bucket', 'mbytes', 'launched_ms', 'backward_end_ms'}]"""
        t = self._tev_done if hasattr(self, '_tev_done') else None
        if not t or t['finish'] is None:
            return []
        end = t['t0'].elapsed_time(t['finish'])
        red = dict(t.get('reduced') or [])
        return [{'bucket': bi, 'mbytes': round(self.buckets[bi]['numel'] * 4 / 1e6, 1) if bi < len(self.buckets) else None,
                 'launched_ms': round(t['t0'].elapsed_time(e), 3), 'backward_end_ms': round(end, 3),
                 # when the compute stream got past its wait for this bucket's collective (>= backward_end_ms by construction:
                 # the waits are issued after backward) - what is above backward_end_ms was NOT hidden behind backward
                 'reduced_ms': round(t['t0'].elapsed_time(red[bi]), 3) if bi in red else None} for bi, e in t['launch']]

    def exposed_ms(self):
        """After a traced step and a device sync: milliseconds between the end of backward and the moment the compute stream
        holds every averaged gradient (waits for the collectives still in flight + the divide / write-back passes) = the part
        of the gradient exchange that did NOT overlap with backward."""
        t = self._tev_done if hasattr(self, '_tev_done') else None
        if not t or t['finish'] is None or t.get('drained') is None:
            return None
        return round(t['finish'].elapsed_time(t['drained']), 3)

    def measure_allreduce(self, reps=5):
        """The step's collectives ALONE (every bucket's all-reduce back to back on the flat buffers, nothing else on the
        device): seconds per sweep from stream events -> algorithm bandwidth bytes / t and the ring 'bus' bandwidth
        2 (N - 1) / N x bytes / t (the per-link figure to hold against xGMI's ~153 GB/s per link).  COLLECTIVE: every rank
        calls it, between steps.  Returns None when the reducer is inactive."""
        if not self.active or not self.buckets or not self.buckets[0]['flat'].is_cuda:
            return None
        for _, h in self._handles:
            h.wait()
        self._handles = []
        nbytes = sum(b['numel'] for b in self.buckets) * 4

        def sweep():
            hs = [dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True) for b in self.buckets]
            for h in hs:
                h.wait()
        sweep()                                   # warm-up (communicator / channel setup)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            sweep()
        e1.record()
        torch.cuda.synchronize()
        sec = e0.elapsed_time(e1) * 1e-3 / reps
        for b in self.buckets:                    # (the sums of sums are garbage: the next step's launches overwrite them)
            b['flat'].zero_()
        n = self.world
        return {'bytes': nbytes, 'buckets': len(self.buckets), 'ms_per_sweep': round(sec * 1e3, 3),
                'algbw_GBs': round(nbytes / sec / 1e9, 2), 'busbw_GBs': round(2 * (n - 1) / n * nbytes / sec / 1e9, 2),
                'sweeps_timed': reps, 'backend': dist.get_backend(self.group)}

    def _launch(self, bi):
        b = self.buckets[bi]
        flat = b['flat']
        if self._tev is not None:
            self._tev['launch'].append((bi, self._event()))
        for p, o in zip(b['params'], b['offsets']):
            if p.grad is None:
                flat[o:o + p.numel()].zero_()
            else:
                flat[o:o + p.numel()].copy_(p.grad.reshape(-1))
        self._handles.append((bi, dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)))

    def _launch_ready(self, force=False):
        while self._next < len(self.buckets) and (force or self._pending[self._next] == 0):
            self._launch(self._next)
            self._next += 1

    def _on_grad(self, p):
        bi = self.where.get(id(p))
        if bi is None:
            # excluded (never-used) parameter got a gradient on this rank.  LOCAL flag only: the layout is collective
            # state and is rebuilt by all ranks together in finish() / resolve()
            self._late = True
            return
        self._pending[bi] -= 1
        if self._discovered:            # step 0 reduces everything in finish(), after the discovery exchange
            self._launch_ready()
        else:
            self._fired.append(p)

    def _used_union(self):
        used = torch.tensor([0 if p.grad is None else 1 for p in self.all_params], dtype=torch.int32,
                            device=self.all_params[0].device)
        dist.all_reduce(used, op=dist.ReduceOp.MAX, group=self.group)
        return used.tolist()

    def _discover(self):
        """Union over ranks of the parameters that got a gradient in this (first) step; rebuild the buckets from them."""
        used = self._used_union()
        # gradient-ready order of rank 0 (position of every parameter in its hook sequence; parameters whose hook did not
        # fire there - used on other ranks only - go last, in registration order)
        pos = {id(p): i for i, p in enumerate(self._fired)}
        n = len(self.all_params)
        order = torch.tensor([pos.get(id(p), n + i) for i, p in enumerate(self.all_params)], dtype=torch.int32,
                             device=self.all_params[0].device)
        dist.broadcast(order, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        order = order.tolist()
        ranked = sorted(range(n), key=lambda i: (order[i], i))
        self.params = [self.all_params[i] for i in ranked if used[i]]
        self.unused = [p for p, u in zip(self.all_params, used) if not u]
        self._fired = []
        self._discovered = True
        self._build_buckets()

    def _drain(self):
        tev = self._tev
        for bi, h in self._handles:
            h.wait()
            if tev is not None:
                tev['reduced'].append((bi, self._event()))
            b = self.buckets[bi]
            b['flat'].div_(self.world)
            for p, o in zip(b['params'], b['offsets']):
                g = b['flat'][o:o + p.numel()].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
        if tev is not None and self._handles:
            tev['drained'] = self._event()
        self._handles = []

    def finish(self, status=None, defer=False):
        """Call after loss.backward(): completes every bucket and writes the averaged gradients back.

        status: optional int32 device vector of the caller; returned MAX-reduced over the ranks (so that e.g. a kernel
        error raised from it is raised on every rank in the same step).  defer=False: the rediscovery flag is read here
        (one device->host read) and acted upon; defer=True: the caller reads the returned word itself - element 0 is the
        flag, the rest is `status` - and MUST call `resolve(word[0])` on every rank before it uses the gradients."""
        if not self.active:
            return None if status is None else torch.cat([status.new_zeros(1), status])
        if not self._discovered:
            self._discover()
        if self._tev is not None:
            self._tev['finish'] = self._event()            # = end of backward on the compute stream
        dev = self.all_params[0].device
        word = torch.zeros(1 + (0 if status is None else status.numel()), dtype=torch.int32, device=dev)
        if self._late:
            word[0] = 1
        if status is not None:
            word[1:] = status.to(device=dev, dtype=torch.int32).reshape(-1)
        # the status word goes LAST: the number of buckets a rank's hooks launched during backward depends on which of its
        # parameters got a gradient (one that lacks it locally holds its bucket back until here), so only "after every
        # bucket" is the same position in the collective sequence on every rank
        self._launch_ready(force=True)
        wh = dist.all_reduce(word, op=dist.ReduceOp.MAX, group=self.group, async_op=True)
        self._drain()
        wh.wait()
        self._late = False
        if self._tev is not None:
            self._tev_done, self._tev = self._tev, None
        self.reset()
        if not defer:
            self.resolve(int(word[0].item()))
        return word

    def resolve(self, flag):
        """Act on the MAX-reduced rediscovery flag of this step (same value on every rank): average the gradients of the
        parameters that were excluded so far but used by some rank in this step, and rebuild the bucket layout."""
        if not self.active or not flag:
            return
        used = self._used_union()
        known = set(id(p) for p in self.params)
        fresh = [p for p, u in zip(self.all_params, used) if u and id(p) not in known]
        for p in fresh:                       # same list, same order on every rank
            g = torch.zeros_like(p) if p.grad is None else p.grad
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
            g.div_(self.world)
            p.grad = g
        self.params = self.params + fresh            # keeps the gradient-ready order; late joiners go last
        self.unused = [p for p in self.all_params if not (id(p) in known or any(p is q for q in fresh))]
        self._build_buckets()
        self.rediscoveries += 1

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
