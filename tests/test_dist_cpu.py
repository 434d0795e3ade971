"""world_size=2 gloo tests (CPU) of the data-parallel path: bucketed/overlapped gradient averaging equals
the mean of per-rank gradients, for a toy module and for the real loss (oracle) evaluated shard-by-shard —
the reference's DataParallel semantics (mean over replicas of per-replica means)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import gvd_amd
from gvd_amd import dist as gdist, synth
from oracle import gvd_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _ParamBag(torch.nn.Module):
    def __init__(self, sd):
        super().__init__()
        self.names = [k for k, v in sd.items() if v.is_floating_point() and 'running' not in k]
        self.p = torch.nn.ParameterList([torch.nn.Parameter(sd[k].clone()) for k in self.names])
        self.rest = {k: v for k, v in sd.items() if k not in self.names}

    def weights(self):
        W = dict(self.rest)
        W.update({k: p for k, p in zip(self.names, self.p)})
        return W


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    gdist.init_from_env('gloo')
    # (1) toy module, tiny buckets -> many collectives, one parameter without gradient
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 4))
    gdist.broadcast_parameters(m)
    red = gdist.GradAllReducer(m, bucket_mb=0.0002)
    x = torch.randn(5, 8, generator=torch.Generator().manual_seed(10 + rank))
    m[2](torch.relu(m[0](x))).pow(2).mean().backward()          # m[3] unused -> no grad
    local = [None if p.grad is None else p.grad.clone() for p in m.parameters()]
    red.finish()
    avg = [None if p.grad is None else p.grad.clone() for p in m.parameters()]
    gathered = [None] * world
    dist.all_gather_object(gathered, local)
    ok = True
    for i, a in enumerate(avg):
        if all(g[i] is None for g in gathered):      # unused on every rank: .grad stays None, as on a single rank
            ok &= a is None
            continue
        want = sum((g[i] if g[i] is not None else torch.zeros_like(a)) for g in gathered) / world
        ok &= torch.allclose(a, want, atol=1e-7)
    # second step: buckets now launch from the hooks, strictly in order; same averages
    for p in m.parameters():
        p.grad = None
    red.reset()
    m[2](torch.relu(m[0](x))).pow(2).mean().backward()
    launched_in_backward = red._next
    red.finish()
    ok &= launched_in_backward == len(red.buckets)     # every bucket was complete before finish()
    ok &= all((a is None and p.grad is None) or torch.allclose(a, p.grad, atol=1e-7) for a, p in zip(avg, m.parameters()))
    # third step, RANK-ASYMMETRIC: only rank 1 runs the so-far excluded m[3].  The rediscovery is a collective decision
    # (MAX-reduced flag in finish()): both ranks rebuild the same layout, m[3]'s gradient is averaged in the SAME step
    # (zeros from rank 0), nothing hangs, and the next step runs on the new layout
    for p in m.parameters():
        p.grad = None
    red.reset()
    h = m[2](torch.relu(m[0](x)))
    (m[3](h) if rank == 1 else h).pow(2).mean().backward()
    local3 = [None if p.grad is None else p.grad.clone() for p in m.parameters()]
    red.finish()
    ok &= red.rediscoveries == 1
    g3 = [None] * world
    dist.all_gather_object(g3, local3)
    for i, p in enumerate(m.parameters()):
        want = sum((g[i] if g[i] is not None else torch.zeros_like(p)) for g in g3) / world
        ok &= p.grad is not None and torch.allclose(p.grad, want, atol=1e-7)
    for p in m.parameters():
        p.grad = None
    red.reset()
    m[3](m[2](torch.relu(m[0](x)))).pow(2).mean().backward()
    ok &= red._next == len(red.buckets)                # the new layout (incl. m[3]) launches from the hooks
    red.finish()
    ok &= red.rediscoveries == 1 and all(p.grad is not None for p in m.parameters())
    # caller status word: MAX over ranks comes back on every rank
    for p in m.parameters():
        p.grad = None
    red.reset()
    m[3](m[2](torch.relu(m[0](x)))).pow(2).mean().backward()
    word = red.finish(status=torch.tensor([rank * 7, 0], dtype=torch.int32), defer=True).tolist()
    red.resolve(word[0])
    ok &= word == [0, 7 * (world - 1), 0]
    # an IN-BUCKET parameter without a gradient on one rank (m[3] is bucketed since the rediscovery; rank 0 does not run
    # it): rank 0's hooks cannot launch m[3]'s bucket (nor any later one) during backward, rank 1's launch all of them.
    # The status word must still sit at the same position of the collective sequence on both ranks (after the last
    # bucket) - a word issued before the forced launches would pair with a bucket on the other rank
    for p in m.parameters():
        p.grad = None
    red.reset()
    h = m[2](torch.relu(m[0](x)))
    (m[3](h) if rank == 1 else h).pow(2).mean().backward()
    in_backward = red._next
    local5 = [None if p.grad is None else p.grad.clone() for p in m.parameters()]
    word = red.finish(status=torch.tensor([3 + rank, 1 - rank], dtype=torch.int32), defer=True).tolist()
    red.resolve(word[0])
    ok &= word == [0, 3 + world - 1, 1]
    counts = [None] * world
    dist.all_gather_object(counts, in_backward)
    ok &= counts[1] == len(red.buckets) and counts[0] < counts[1]      # the two ranks really launched differently
    g5 = [None] * world
    dist.all_gather_object(g5, local5)
    for i, p in enumerate(m.parameters()):
        want = sum((g[i] if g[i] is not None else torch.zeros_like(p)) for g in g5) / world
        ok &= p.grad is not None and torch.allclose(p.grad, want, atol=1e-7)
    # (2) the real loss on this rank's shard of a batch (oracle), averaged grads == mean of shard grads
    opt = gvd_amd.opts.default_opt(vocab_size=120, t_attn_size=6)
    sd = synth.init_state_dict(opt, seed=4)
    full = synth.trim_to_batch(synth.make_inputs(opt, 2 * world, seed=4, train=True))
    mine = synth.shard(full, rank, world)
    bag = _ParamBag(sd)
    red2 = gdist.GradAllReducer(bag, bucket_mb=8)
    lm, a2, gl, cl, _ = O.forward_train(bag.weights(), opt, *[mine[k] for k in synth.FORWARD_ORDER])
    (lm + 0.05 * a2 + 0.1 * cl).backward()
    local_losses = torch.stack([lm, a2, gl, cl]).detach()
    local_g = {n: (None if p.grad is None else p.grad.clone()) for n, p in zip(bag.names, bag.p)}
    red2.finish()
    out = dict(rank=rank, ok_toy=bool(ok), losses=local_losses,
               local=local_g, avg={n: (None if p.grad is None else p.grad.clone()) for n, p in zip(bag.names, bag.p)})
    torch.save(out, os.path.join(outdir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gradient_averaging(tmp_path):
    world = 2
    ctx = mp.get_context('spawn')
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=500)
        assert p.exitcode == 0
    res = [torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r), weights_only=False) for r in range(world)]
    res.sort(key=lambda r: r['rank'])
    assert all(r['ok_toy'] for r in res)
    names = list(res[0]['avg'])
    for n in names:
        l0, l1 = res[0]['local'][n], res[1]['local'][n]
        a = res[0]['avg'][n]
        if l0 is None and l1 is None:
            assert a is None and res[1]['avg'][n] is None, n          # never-used parameters keep .grad None
            continue
        want = ((l0 if l0 is not None else torch.zeros_like(a)) + (l1 if l1 is not None else torch.zeros_like(a))) / 2
        assert torch.allclose(a, want, rtol=1e-6, atol=1e-8), n
        assert torch.equal(res[0]['avg'][n], res[1]['avg'][n]), n        # replicas stay in lock-step
    # the two shards really are different work
    assert not torch.allclose(res[0]['losses'], res[1]['losses'])
