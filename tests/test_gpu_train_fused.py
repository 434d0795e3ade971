"""-m gpu: the fused elementwise kernels of the training step (csrc/train_fused.hip, the dropout forms of the residual
LayerNorm in csrc/rowwise.hip) and the optimiser kernels (csrc/optim.hip) against plain torch formulations of the same
arithmetic (fp64 where a gradient is compared)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gvd_amd import ops
from gvd_amd.optim import ClipAdam

pytestmark = pytest.mark.gpu


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _mask(shape, p, seed):
    """The keep mask (0 / 1) the library draws for (seed, p) over a contiguous tensor of this shape."""
    return (ops.dropout_rows(torch.ones(shape, device='cuda'), p, seed) != 0).float()


@pytest.mark.parametrize('shape,p', [((1000, 2048), 0.5), ((64, 300), 0.2), ((3, 4), 0.5), ((257, 1024), 0.9)])
def test_dropout_rows_statistics_and_determinism(shape, p):
    g = _g(7)
    x = (torch.randn(*shape, generator=g) + 3.0).cuda()
    y = ops.dropout_rows(x, p, 99)
    keep = y != 0
    # kept elements are scaled by exactly 1 / (1 - p), dropped ones are exactly 0
    np.testing.assert_allclose(y[keep].cpu().numpy(), (x[keep] * (1.0 / (1.0 - p))).cpu().numpy(), rtol=1e-6)
    n = x.numel()
    if n >= 10000:
        frac = float(keep.float().mean())
        assert abs(frac - (1 - p)) < 5 * np.sqrt(p * (1 - p) / n), frac
        # no structure along rows / columns: every row and column keeps roughly its share
        assert float(keep.float().mean(1).min()) > (1 - p) - 0.25 and float(keep.float().mean(0).min()) > (1 - p) - 0.25
    assert torch.equal(y, ops.dropout_rows(x, p, 99))                       # same seed -> same mask
    assert not torch.equal(y, ops.dropout_rows(x, p, 100))                  # another seed -> another mask
    z = x.clone()
    assert ops.dropout_(z, p, 99) is z and torch.equal(z, y)                # in place == out of place
    assert torch.equal(ops.dropout_rows(x, 0.0, 5), x)


@pytest.mark.parametrize('M,N,p', [(1000, 2048, 0.5), (130, 300, 0.2), (64, 1024, 0.0), (7, 512, 0.5), (4099, 2784, 0.5)])
def test_relu_dropout_bwd_colsum(M, N, p):
    g = _g(M + N)
    z = torch.randn(M, N, generator=g).cuda()
    y = torch.relu(z)
    if p > 0:
        ops.dropout_(y, p, 31337)
    dy = torch.randn(M, N, generator=g).cuda()
    dz, db = ops.relu_dropout_bwd(dy, y, p)
    want = torch.where(y > 0, dy * (1.0 / (1.0 - p)), torch.zeros_like(dy))
    assert torch.equal(dz, want)
    ref_db = want.double().sum(0)
    assert float((db.double() - ref_db).abs().max()) <= 1e-5 * max(1.0, float(ref_db.abs().max()))
    dz2, db2 = ops.relu_dropout_bwd(dy, y, p)
    assert torch.equal(db, db2)                                              # ordered partial sums


@pytest.mark.parametrize('p', [0.0, 0.5])
def test_linear_relu_dropout_autograd(p):
    """ops.linear(x, w, b, act=1, p_drop): forward y = dropout(relu(x W^T + b)) with the library's mask, backward through
    the fused mask + bias-gradient pass and the K-strided products, against autograd (fp64) through the same expression."""
    g = _g(11)
    M, K, N = 4096, 1024, 1024           # 256 output tiles of dX: the pipelined K-strided MFMA kernel
    x = torch.randn(M, K, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(N, K, generator=g) * 0.1).cuda().requires_grad_(True)
    b = (torch.randn(N, generator=g) * 0.1).cuda().requires_grad_(True)
    dy = torch.randn(M, N, generator=g).cuda()
    torch.manual_seed(123)
    y = ops.linear(x, w, b, 1, p)
    y.backward(dy)
    got = [t.grad.clone() for t in (x, w, b)]
    torch.manual_seed(123)
    mask = _mask((M, N), p, ops.draw_seed()).double() if p > 0 else torch.ones(M, N, device='cuda', dtype=torch.float64)
    x64, w64, b64 = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    ref = torch.relu(x64 @ w64.t() + b64) * mask * (1.0 / (1.0 - p))
    ref.backward(dy.double())
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.detach().float().cpu().numpy(), rtol=1e-4, atol=1e-4)   # (fp32 K = 1024 sums)
    for a, r, name in zip(got, (x64.grad, w64.grad, b64.grad), ('dx', 'dw', 'db')):
        err = float((a.double() - r).abs().max())
        assert err <= 2e-5 * max(1.0, float(r.abs().max())), (name, err)
    if p > 0:
        frac = float((y == 0).float().mean())
        assert 0.70 < frac < 0.80       # relu zeroes ~half, dropout half of the rest


@pytest.mark.parametrize('rows,p', [(1000, 0.1), (130, 0.5), (7, 0.3)])
def test_add_layernorm_dropout_fwd_bwd(rows, p):
    """LayerNorm_unbiased(x + dropout(y)) (transformer.py:79-88 in training mode): the fused kernels against the module's
    elementwise formulation applied to the explicitly dropped branch (the mask the library draws for the same seed)."""
    from gvd_amd.att_model import _EncLayerNorm
    g = _g(5 + rows)
    D = 1024
    ln = _EncLayerNorm(D).cuda()
    with torch.no_grad():
        ln.gamma.copy_(torch.randn(D, generator=g) * 0.5 + 1)
        ln.beta.copy_(torch.randn(D, generator=g) * 0.1)
    x = torch.randn(rows, D, generator=g).cuda().requires_grad_(True)
    y = (torch.randn(rows, D, generator=g) * 0.7).cuda().requires_grad_(True)
    dout = torch.randn(rows, D, generator=g).cuda()
    torch.manual_seed(77)
    out = ops.add_layernorm(x, y, ln.gamma, ln.beta, ln.eps, p)
    out.backward(dout)
    got = [t.grad.clone() for t in (x, y, ln.gamma, ln.beta)]
    torch.manual_seed(77)
    seed = ops.draw_seed()
    mask = _mask((rows, D), p, seed).double()
    assert 0 < float(mask.mean()) < 1
    ln64 = _EncLayerNorm(D).cuda().double()
    with torch.no_grad():
        ln64.gamma.copy_(ln.gamma.double()); ln64.beta.copy_(ln.beta.double())
    x64, y64 = x.detach().double().requires_grad_(True), y.detach().double().requires_grad_(True)
    ref = ln64(x64 + y64 * mask * (1.0 / (1.0 - p)))
    ref.backward(dout.double())
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().float().cpu().numpy(), rtol=2e-5, atol=2e-5)
    for a, r, name in zip(got, (x64.grad, y64.grad, ln64.gamma.grad, ln64.beta.grad), ('dx', 'dy', 'dgamma', 'dbeta')):
        err = float((a.double() - r).abs().max())
        assert err <= 2e-5 * max(1.0, float(r.abs().max())), (name, err)
    assert torch.equal(got[1] != 0, (mask != 0) & (got[0] != 0))            # the branch gradient is masked like the forward
    # no-grad form: the same values
    torch.manual_seed(77)
    with torch.no_grad():
        out2 = ops.add_layernorm(x.detach(), y.detach(), ln.gamma, ln.beta, ln.eps, p)
    np.testing.assert_allclose(out2.cpu().numpy(), out.detach().cpu().numpy(), rtol=1e-6, atol=1e-6)


def test_sum_chunks_pair_and_second_dh_addend():
    g = _g(3)
    B, A, H = 37, 512, 1024
    a, r = torch.randn(B, 1, A, generator=g).cuda(), torch.randn(B, 21, A, generator=g).cuda()
    out = torch.full((B, 2 * A + 8), float('nan'), device='cuda')[:, :2 * A]        # a strided view
    ops.sum_chunks_pair(a, r, out)
    np.testing.assert_allclose(out[:, :A].cpu().numpy(), a.sum(1).cpu().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(out[:, A:].cpu().numpy(), r.double().sum(1).float().cpu().numpy(), rtol=1e-5, atol=1e-5)
    gates = torch.cat([torch.rand(B, H, generator=g), torch.rand(B, H, generator=g),
                       torch.rand(B, H, generator=g) * 2 - 1, torch.rand(B, H, generator=g)], 1).cuda()
    dh, dh2, dc, cp, cn = (torch.randn(B, H, generator=g).cuda() for _ in range(5))
    dg, dcp = ops.lstm_cell_bwd(dh, dc, gates, cp, cn, dh2=dh2)
    dg_ref, dcp_ref = ops.lstm_cell_bwd(dh + dh2, dc, gates, cp, cn)
    assert torch.equal(dg, dg_ref) and torch.equal(dcp, dcp_ref)


@pytest.mark.parametrize('wd', [0.0, 0.01])
def test_clip_adam_matches_clip_grad_norm_plus_torch_adam(wd):
    """main.py:265-266 (clip_grad_norm_(0.1) + Adam.step) on the library's kernels vs torch's: 4 steps over tensors of
    awkward sizes (1 element, 433, not a multiple of 4, > one chunk, > 32 tensors = two launches, one without gradient)."""
    g = _g(17)
    sizes = [(1,), (433,), (5000, 1027), (3, 5), (2048, 2048)] + [(64, 33)] * 30 + [(7,)]
    ps_a = [torch.nn.Parameter((torch.randn(*s, generator=g) * 0.1).cuda()) for s in sizes]
    ps_b = [torch.nn.Parameter(p.detach().clone()) for p in ps_a]

    def groups(ps):
        return [{'params': ps[:20], 'lr': 5e-4, 'weight_decay': wd, 'betas': (0.8, 0.999)},
                {'params': ps[20:], 'lr': 5e-5, 'weight_decay': wd, 'betas': (0.8, 0.999)}]
    own, ref = ClipAdam(groups(ps_a)), torch.optim.Adam(groups(ps_b))
    for step in range(4):
        for i, (pa, pb) in enumerate(zip(ps_a, ps_b)):
            if i == len(sizes) - 1:
                pa.grad = pb.grad = None                   # never used (core.i2h_2 of the model): must not move
                continue
            gr = (torch.randn(*pa.shape, generator=g) * (10.0 if step < 2 else 1e-6)).cuda()     # clipped / not clipped
            pa.grad, pb.grad = gr.clone(), gr.clone()
        norm = own.step_clipped(0.1)
        ref_norm = torch.nn.utils.clip_grad_norm_(ps_b, 0.1)
        ref.step()
        assert abs(float(norm) - float(ref_norm)) <= 1e-5 * float(ref_norm)
        for i, (pa, pb) in enumerate(zip(ps_a, ps_b)):
            err = float((pa.detach() - pb.detach()).abs().max())
            assert err <= 2e-6 + 1e-4 * 5e-4, (step, i, err)     # an update is at most ~lr; fp32 rounding of the two forms
    for pa, pb in zip(ps_a, ps_b):
        if pb in ref.state:
            np.testing.assert_allclose(own.state[pa]['exp_avg'].cpu().numpy(), ref.state[pb]['exp_avg'].cpu().numpy(),
                                       rtol=1e-4, atol=1e-7)      # (g + wd p cancels to ~1e-6 in places)
            np.testing.assert_allclose(own.state[pa]['exp_avg_sq'].cpu().numpy(), ref.state[pb]['exp_avg_sq'].cpu().numpy(),
                                       rtol=1e-4, atol=1e-12)
            assert float(own.state[pa]['step']) == float(ref.state[pb]['step']) == 4
    assert len(own.state[ps_a[-1]]) == 0
    # state_dict interchange with torch.optim.Adam (what the reference's checkpoints would hold)
    sd = own.state_dict()
    other = torch.optim.Adam(groups(ps_b))
    other.load_state_dict(sd)
    own2 = ClipAdam(groups(ps_a))
    own2.load_state_dict(ref.state_dict())
    assert float(own2.state[ps_a[0]]['step']) == 4


def test_trainer_step_equals_clip_grad_norm_plus_torch_adam():
    """train.Trainer (optim.ClipAdam: clip factor and Adam on the library's kernels) against main.py:263-266 spelled out with
    torch - loss.backward(), clip_grad_norm_(0.1), torch.optim.Adam over the same two learning-rate groups - from the same
    state on the same batch (eval-mode arithmetic: identical gradients): ONE step moves every parameter alike; the second
    step's losses agree (its parameter updates are not compared element by element: a 1e-10 parameter difference can flip a
    ReLU unit that sits on its boundary, and Adam turns the changed gradient entries into O(lr) differences)."""
    import gvd_amd
    from gvd_amd import att_model, synth, train
    opt = gvd_amd.opts.default_opt(vocab_size=1000, t_attn_size=10)
    sd = synth.init_state_dict(opt, seed=3, profile='trained_like')
    inp = synth.trim_to_batch(synth.make_inputs(opt, 4, seed=3, train=True))
    args = synth.as_args(inp, 'cuda')

    def fresh():
        model = att_model.TopDownModel(opt)
        model.load_state_dict(sd)
        return model.cuda().eval()
    model = fresh()
    tr = train.Trainer(model, opt)
    assert type(tr.optimizer).__name__ == 'ClipAdam'
    l1 = tr.step(args).cpu()
    after1 = {n: p.detach().clone() for n, p in model.named_parameters()}
    l2 = tr.step(args).cpu()
    norm = tr.last_grad_norm

    ref = fresh()
    groups = [{'params': list(g['params']), 'lr': g['lr'], 'weight_decay': g['weight_decay'], 'betas': g['betas']}
              for g in train.build_optimizer(ref, opt).param_groups]
    adam = torch.optim.Adam(groups)
    steps = []
    for _ in range(2):
        ref.zero_grad(set_to_none=True)
        losses = ref(*args, 'MLE')
        train.combine_losses(losses, opt).backward()
        ref_norm = torch.nn.utils.clip_grad_norm_(ref.parameters(), opt.grad_clip)
        adam.step()
        steps.append((torch.cat([l.detach() for l in losses]).cpu(), float(ref_norm),
                      {n: p.detach().clone() for n, p in ref.named_parameters()}))
    assert torch.equal(l1, steps[0][0])
    np.testing.assert_allclose(l2.numpy(), steps[1][0].numpy(), rtol=1e-5, atol=1e-6)
    assert abs(norm - steps[1][1]) <= 1e-4 * steps[1][1]
    for n in after1:
        err = float((after1[n] - steps[0][2][n]).abs().max())
        assert err <= 2e-6, (n, err)


def test_clip_adam_device_side_skip_flag():
    """step_clipped(skip=word): a raised word leaves parameters and moments untouched (the host then rolls the step
    counters back), a clear word steps exactly like an unpredicated call."""
    g = _g(29)
    ps = [torch.nn.Parameter(torch.randn(n, generator=g).cuda()) for n in (5, 40000, 433)]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    a, b = ClipAdam([{'params': ps, 'lr': 1e-3}]), ClipAdam([{'params': ref, 'lr': 1e-3}])
    for p, r in zip(ps, ref):
        gr = torch.randn(p.shape, generator=g).cuda()
        p.grad, r.grad = gr.clone(), gr.clone()
    before = [p.detach().clone() for p in ps]
    raised = torch.tensor([0, 3, 0], dtype=torch.int32, device='cuda')
    norm = a.step_clipped(0.1, skip=raised)
    assert all(torch.equal(p.detach(), q) for p, q in zip(ps, before))
    assert all(float(a.state[p]['exp_avg'].abs().max()) == 0 and float(a.state[p]['exp_avg_sq'].abs().max()) == 0 for p in ps)
    assert all(float(a.state[p]['step']) == 1 for p in ps)
    a.rollback_step_counts()
    assert all(float(a.state[p]['step']) == 0 for p in ps)
    clear = torch.zeros(3, dtype=torch.int32, device='cuda')
    norm2 = a.step_clipped(0.1, skip=clear)
    norm3 = b.step_clipped(0.1)
    assert float(norm) == float(norm2) == float(norm3)
    assert not any(torch.equal(p.detach(), q) for p, q in zip(ps, before))
    for p, r in zip(ps, ref):
        assert torch.equal(p.detach(), r.detach())
        assert torch.equal(a.state[p]['exp_avg'], b.state[r]['exp_avg']) and float(a.state[p]['step']) == 1
