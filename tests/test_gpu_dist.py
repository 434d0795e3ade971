"""-m gpu: the data-parallel training path with the REAL HIP model — two gloo ranks share cuda:0 (the GPU box has one
device; the collective goes through gloo, everything else is the production path: train.Trainer, dist.GradAllReducer
hooks/buckets, the HIP forward + hand-scheduled BPTT).  Checks the reference's DataParallel semantics (SURVEY.md §8e,
main.py:239-255): gradients after the all-reduce = mean over ranks of the per-shard gradients, replicas stay
bit-identical after the optimizer step, bucket launches happen from the backward hooks in bucket order."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    import gvd_amd
    from gvd_amd import att_model, dist as gdist, synth, train
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    opt = gvd_amd.opts.default_opt(vocab_size=400, t_attn_size=8, w_att2=0.05, w_grd=0.3, w_cls=0.1)
    sd = synth.init_state_dict(opt, seed=6, profile='trained_like')
    model = att_model.TopDownModel(opt)
    model.load_state_dict(sd)
    model = model.cuda().eval()               # eval arithmetic: dropout off, BN running stats -> comparable across ranks
    gdist.broadcast_parameters(model)
    full = synth.trim_to_batch(synth.make_inputs(opt, 2 * world, seed=6, train=True))
    mine = synth.shard(full, rank, world)
    tr = train.Trainer(model, opt, bucket_mb=16)
    grads = []
    launched = []
    for step in range(2):                     # step 0 = discovery pass, step 1 = hooks launch the buckets mid-backward
        model.zero_grad(set_to_none=True)
        tr.reducer.reset()
        losses = model(*synth.as_args(mine, 'cuda'), 'MLE')
        loss = train.combine_losses(losses, opt)
        loss.backward()
        local = {n: (None if p.grad is None else p.grad.detach().clone().cpu()) for n, p in model.named_parameters()}
        launched.append(tr.reducer._next)
        tr.reducer.finish()
        avg = {n: (None if p.grad is None else p.grad.detach().clone().cpu()) for n, p in model.named_parameters()}
        grads.append((local, avg))
    # one real optimizer step through the Trainer, then compare replicas
    tr.step(synth.as_args(mine, 'cuda'))
    params = {n: p.detach().cpu() for n, p in model.named_parameters()}
    torch.save(dict(rank=rank, grads=grads, params=params, launched=launched, nbuckets=len(tr.reducer.buckets),
                    losses=torch.cat([l.detach().cpu() for l in losses])), os.path.join(outdir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_real_model_gradient_averaging(tmp_path):
    world = 2
    ctx = mp.get_context('spawn')
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=800)
        assert p.exitcode == 0
    res = sorted([torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r), weights_only=False) for r in range(world)],
                 key=lambda r: r['rank'])
    assert not torch.allclose(res[0]['losses'], res[1]['losses'])           # the two shards are different work
    for step in range(2):
        (l0, a0), (l1, a1) = res[0]['grads'][step], res[1]['grads'][step]
        for n in a0:
            if l0[n] is None and l1[n] is None:
                assert a0[n] is None and a1[n] is None, n                    # unused core.i2h_2 / h2h_2 keep .grad None
                continue
            z = torch.zeros_like(a0[n])
            want = ((l0[n] if l0[n] is not None else z) + (l1[n] if l1[n] is not None else z)) / 2
            assert torch.allclose(a0[n], want, rtol=1e-6, atol=1e-9), n
            assert torch.equal(a0[n], a1[n]), n
    # second step: buckets were launched from the hooks during backward (all but possibly the last complete)
    assert res[0]['nbuckets'] >= 2 and res[0]['launched'][0] == 0 and res[0]['launched'][1] >= res[0]['nbuckets'] - 1
    for n, p in res[0]['params'].items():
        assert torch.equal(p, res[1]['params'][n]), n                        # replicas stay in lock-step
