"""-m gpu: the streaming products (csrc/stream_mm.hip), the small-M dX launch of the BPTT (csrc/gemm_dxs.hip) and the
row / column kernels of the training step (csrc/train_rows.hip) against fp64 / autograd formulations of the reference ops
they replace: `_grounder` (model.py:243-280) forward and both gradients, alpha^T d_ctx, nn.LSTMCell / nn.Linear input
gradients, utils.LMCriterion's reductions (utils.py:122-152), nn.BatchNorm1d in train mode (model.py:114,397)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import gvd_amd
from gvd_amd import ops
from oracle import gvd_oracle as O

pytestmark = pytest.mark.gpu


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _close(got, want64, rel=2e-5, what=''):
    """|got - want| <= rel * max|want| (+ tiny): fp32 accumulation of up to a few thousand terms in another order."""
    got, want64 = got.double().cpu(), want64.double().cpu()
    scale = float(want64.abs().max()) + 1e-30
    err = float((got - want64).abs().max())
    assert err <= rel * scale + 1e-7, '%s: max abs err %.3e vs scale %.3e' % (what, err, scale)


@pytest.mark.parametrize('B,M,R,K,mask_dim', [(3, 20, 1000, 2048, 3), (2, 7, 203, 2048, 2), (1, 32, 64, 512, 3),
                                              (5, 1, 130, 128, 0), (2, 17, 129, 96, 3), (64, 20, 1000, 2048, 3)])
def test_grounder_stream_forward(B, M, R, K, mask_dim):
    g = _g(B * 1000 + M)
    xt = torch.relu(torch.randn(B, M, K, generator=g))
    feats = torch.relu(torch.randn(B, R, K, generator=g))
    mbias = torch.randn(B, M, generator=g)
    rowbias = torch.randn(B, M, R, generator=g)
    mask = None
    if mask_dim == 2:
        mask = (torch.rand(B, R, generator=g) < 0.3).to(torch.uint8)
    elif mask_dim == 3:
        mask = (torch.rand(B, M, R + 1, generator=g) < 0.3).to(torch.uint8)[:, :, 1:]      # (an offset view, like frm_masks[:, :, 1:])
    dmask = None
    if mask is not None:       # the 3-D mask goes in as an OFFSET VIEW on the device too (row stride R + 1, 1-byte misaligned rows)
        dmask = mask.cuda() if mask_dim == 2 else mask._base.cuda()[:, :, 1:]
        assert mask_dim == 2 or not dmask.is_contiguous()
    out = ops.grounder_stream(xt.cuda(), feats.cuda(), dmask, mbias.cuda(), rowbias.cuda())
    want = torch.bmm(xt.double(), feats.double().transpose(1, 2)) + mbias.double().unsqueeze(2) + rowbias.double()
    if mask is not None:
        mm = mask.bool() if mask.dim() == 3 else mask.bool().unsqueeze(1)
        assert torch.equal(out.cpu()[mm.expand_as(out)], torch.full((int(mm.expand_as(out).sum()),), O.MIN_VALUE))
        want = want.masked_fill(mm, 0.0)
        out = out.cpu().masked_fill(mm, 0.0)
    _close(out, want, what='grounder forward')
    # the reference's own formulation (model.py:262-278) through the oracle, fp32: bias = class bias + region-attention logits
    if B <= 3 and mask is not None:
        m3 = mask if mask.dim() == 3 else mask.unsqueeze(1).expand(B, M, R)
        ref = O.grounder_dot(xt, feats, m3, mbias.unsqueeze(2) + rowbias)
        got = ops.grounder_stream(xt.cuda(), feats.cuda(), mask.cuda(), mbias.cuda(), rowbias.cuda()).cpu()
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-4, atol=2e-3)


def test_grounder_stream_equals_the_batched_gemm_path():
    """Same product, two kernels (the batched 32 x 128 MFMA GEMM that served the grounder before, the streaming kernel now):
    both are fp32 FMA chains over k; they differ in the order the biases are added in the epilogue."""
    g = _g(5)
    B, M, R, K = 4, 20, 1000, 2048
    xt, feats = torch.randn(B, M, K, generator=g).cuda(), torch.randn(B, R, K, generator=g).cuda()
    mask = (torch.rand(B, M, R, generator=g) < 0.2).to(torch.uint8).cuda()
    mb, rb = torch.randn(B, M, generator=g).cuda(), torch.randn(B, M, R, generator=g).cuda()
    a = ops.grounder_stream(xt, feats, mask, mb, rb)
    b = ops.grounder_dot(xt, feats, mask, mb, rb)
    err = float((a - b).abs().max())
    assert err <= 2e-4 * float(b.abs().max()), err        # (the epilogues add the biases in a different order: not bitwise)


@pytest.mark.parametrize('B,M,R,N', [(3, 20, 1000, 2048), (2, 7, 203, 1024), (1, 32, 64, 128), (4, 1, 37, 256),
                                     (2, 20, 10, 1024), (64, 20, 1000, 2048)])
def test_rows_contract_and_rank_update(B, M, R, N):
    g = _g(B * 100 + R)
    S = torch.randn(B, M, R, generator=g)
    Fm = torch.randn(B, R, N, generator=g)
    X = torch.randn(B, M, N, generator=g)
    mask = (torch.rand(B, M, R, generator=g) < 0.3).to(torch.uint8)
    Sm = S.double().masked_fill(mask.bool(), 0.0)
    for use_mask in (False, True):
        Sd = Sm if use_mask else S.double()
        mk = mask.cuda() if use_mask else None
        out = ops.rows_contract(S.cuda(), Fm.cuda(), mk)
        _close(out, torch.bmm(Sd, Fm.double()), what='rows_contract')
        out2 = ops.rank_update(S.cuda(), X.cuda(), mk)
        _close(out2, torch.bmm(Sd.transpose(1, 2), X.double()), what='rank_update')
    # the masked copy + row sums + transposed copy of the grounder's backward, and rows_contract reading S through the latter
    dm, rs, dmt = ops.masked_copy_rowsum(S.cuda(), mask.cuda(), want_sum=True, want_t=True)
    assert torch.equal(dm.cpu(), S.masked_fill(mask.bool(), 0.0))
    np.testing.assert_allclose(rs.cpu().numpy(), Sm.sum(-1).numpy(), rtol=1e-4, atol=1e-4)
    assert torch.equal(dmt[:, :, :M].cpu(), dm.cpu().transpose(1, 2)) and float(dmt[:, :, M:].abs().max() if M < 32 else 0.0) == 0.0
    _close(ops.rows_contract(dm, Fm.cuda(), S_t=dmt), torch.bmm(Sm, Fm.double()), what='rows_contract via S_t')
    # strided few-row operand (the BPTT hands in dX_all[:, :, :H].transpose(0, 1))
    big = torch.randn(M, B, 2 * N, generator=g).cuda()
    Xv = big[:, :, :N].transpose(0, 1)
    _close(ops.rank_update(S.cuda(), Xv), torch.bmm(S.double().transpose(1, 2), Xv.double().cpu()), what='rank_update strided')
    # 2-D mask (one row per sample)
    m2 = (torch.rand(B, R, generator=g) < 0.4).to(torch.uint8)
    Sd = S.double().masked_fill(m2.bool().unsqueeze(1), 0.0)
    _close(ops.rows_contract(S.cuda(), Fm.cuda(), m2.cuda()), torch.bmm(Sd, Fm.double()), what='rows_contract 2-D mask')


def test_grounder_autograd_matches_torch():
    """ops.grounder under autograd (streaming forward, masked copy + row sums, rows_contract, rank_update) against autograd
    through the reference formulation (bmm + bias + masked_fill, model.py:262-278)."""
    g = _g(11)
    B, M, R, K = 3, 20, 400, 2048
    xt = torch.relu(torch.randn(B, M, K, generator=g)) * 0.1
    feats = torch.relu(torch.randn(B, R, K, generator=g)) * 0.1
    mb, rb = torch.randn(B, M, generator=g), torch.randn(B, M, R, generator=g)
    mask = (torch.rand(B, M, R, generator=g) < 0.3).to(torch.uint8)
    w = torch.randn(B, M, R, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (xt, feats, mb, rb)]
    ref = torch.bmm(leaves[0], leaves[1].transpose(1, 2)) + leaves[2].unsqueeze(2) + leaves[3]
    ref = ref.masked_fill(mask.bool(), O.MIN_VALUE)
    (ref * w).sum().backward()
    dev = [t.clone().cuda().requires_grad_(True) for t in (xt, feats, mb, rb)]
    out = ops.grounder(dev[0], dev[1], mask.cuda(), mbias=dev[2], rowbias=dev[3])
    (out * w.cuda()).sum().backward()
    keep = ~mask.bool()
    _close(out.detach().cpu()[keep], ref.detach().double()[keep], what='forward')
    for a, b, n in zip(dev, leaves, ('xt', 'feats', 'mbias', 'rowbias')):
        _close(a.grad, b.grad.double(), rel=5e-5, what='grad ' + n)


@pytest.mark.parametrize('M', [64, 32, 20, 1, 100])
def test_dx_products_one_launch(M):
    """The three gate products of a BPTT step + the addend form, against fp64; two launches give the same bits (the split-K
    partials are added in slice order whichever workgroup arrives last)."""
    g = _g(M)
    H = 1024
    dg_l, dg_a = torch.randn(M, 4 * H, generator=g).cuda(), torch.randn(M, 4 * H, generator=g).cuda()
    w_ih, w_hh, w_ahh = (torch.randn(4 * H, n, generator=g).cuda() * 0.05 for n in (2 * H, H, H))
    outs = []
    for _ in range(2):
        dX = torch.full((M, 2 * H), float('nan'), device='cuda')
        dh, dha = torch.full((M, H), float('nan'), device='cuda'), torch.full((M, H), float('nan'), device='cuda')
        ops.dx_products([dict(A=dg_l, W=w_ih, out=dX), dict(A=dg_l, W=w_hh, out=dh), dict(A=dg_a, W=w_ahh, out=dha)], M)
        outs.append((dX, dh, dha))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    for got, A, W in zip(outs[0], (dg_l, dg_l, dg_a), (w_ih, w_hh, w_ahh)):
        _close(got, A.double().cpu() @ W.double().cpu(), what='dx product')
    # addend + a column-block view of a wider weight + a strided output view (d h_att = dX[:, H:] + dq12 W_stack)
    dq = torch.randn(M, 1024, generator=g).cuda()
    wide = torch.randn(1024, 1536, generator=g).cuda() * 0.05
    buf = torch.zeros(M, 2 * H, device='cuda')
    add = torch.randn(M, 2 * H, generator=g).cuda()
    ops.dx_products([dict(A=dq, W=wide[:, 512:], out=buf[:, H:], addend=add[:, H:])], M)
    _close(buf[:, H:], (dq.double() @ wide[:, 512:].double() + add[:, H:].double()).cpu(), what='addend form')
    assert float(buf[:, :H].abs().max()) == 0.0
    # post-loop shape: [Lc B, 4H] x [4H, 512]
    if M == 64:
        A = torch.randn(1280, 4 * H, generator=g).cuda()
        o = torch.empty(1280, 512, device='cuda')
        ops.dx_products([dict(A=A, W=w_ih[:, 1024:1536], out=o)], 1280)
        _close(o, (A.double() @ w_ih[:, 1024:1536].double()).cpu(), what='post-loop dx')


def test_dx_products_share_one_workspace_across_shapes():
    """One workspace serves launches of different shapes (every BPTT step's M = 64 products, then the [Lc B, .] post-loop ones):
    the tile counters must stay zero and never land on bytes another shape used for partial sums.  Largest shape FIRST, so that
    the workspace is never regrown (a fresh, zeroed one would hide the bug the full suite found at B = 64)."""
    g = _g(77)
    H = 1024
    A_big, W_big = torch.randn(1280, 4 * H, generator=g).cuda(), torch.randn(4 * H, 512, generator=g).cuda() * 0.05
    A64 = torch.randn(64, 4 * H, generator=g).cuda()
    W3 = [torch.randn(4 * H, n, generator=g).cuda() * 0.05 for n in (2 * H, H, H)]
    want_big = (A_big.double() @ W_big.double()).cpu()
    want3 = [(A64.double() @ w.double()).cpu() for w in W3]
    for rnd in range(3):
        o = torch.empty(1280, 512, device='cuda')
        ops.dx_products([dict(A=A_big, W=W_big, out=o)], 1280)
        _close(o, want_big, what='post-loop shape, round %d' % rnd)
        outs = [torch.empty(64, w.shape[1], device='cuda') for w in W3]
        ops.dx_products([dict(A=A64, W=w, out=t) for w, t in zip(W3, outs)], 64)
        for t, wnt in zip(outs, want3):
            _close(t, wnt, what='step shape, round %d' % rnd)
        o32 = torch.empty(32, 512, device='cuda')
        ops.dx_products([dict(A=A_big[:32], W=W_big, out=o32)], 32)
        _close(o32, want_big[:32], what='M = 32, round %d' % rnd)


@pytest.mark.parametrize('M,N', [(1280, 5000), (80, 1000), (1280, 1024), (37, 200)])
def test_gemm_dx_small_with_a_contraction_tail(M, N):
    """ops.gemm_dx_small: dX = dY W for the token loop's [Lc B, .] shapes, the contraction cut into its leading multiple of 128
    (in place) + a zero-padded tail (vocabulary head: 5000 = 39 x 128 + 8) that enters the main launch as its addend; and
    through the nn.Linear autograd Function (ops.linear), which takes this path when the product has fewer than 256 tiles."""
    g = _g(M + N)
    dY = torch.randn(M, N, generator=g).cuda()
    W = (torch.randn(N, 1024, generator=g) * 0.05).cuda()
    got = ops.gemm_dx_small(dY, W)
    assert got is not None
    _close(got, (dY.double() @ W.double()).cpu(), what='dX [%d,%d]x[%d,1024]' % (M, N, N))
    x = torch.randn(M, 1024, generator=g).cuda().requires_grad_(True)
    w = W.clone().requires_grad_(True)
    y = ops.linear(x, w)
    y.backward(dY)
    _close(x.grad, (dY.double() @ W.double()).cpu(), what='ops.linear d x')
    # a row pitch that is not a multiple of 16 bytes (odd vocabulary size): zero-padded copies of both operands, still on the
    # kernel - ops.linear's backward (dX and dW) never reaches the library (the suite runs under GVD_STRICT)
    n0 = ops.library_call_count()
    _close(ops.gemm_dx_small(dY[:, :N - 1].contiguous(), W[:N - 1]), (dY[:, :N - 1].double() @ W[:N - 1].double()).cpu(),
           what='dX, odd width')
    x2 = x.detach().clone().requires_grad_(True)
    w2 = w[:N - 1].detach().contiguous().requires_grad_(True)
    ops.linear(x2, w2).backward(dY[:, :N - 1].contiguous())
    _close(x2.grad, (dY[:, :N - 1].double() @ W[:N - 1].double()).cpu(), what='ops.linear d x, odd width')
    _close(w2.grad, (dY[:, :N - 1].double().t() @ x.detach().double()).cpu(), what='ops.linear d w, odd width')
    assert ops.library_call_count() == n0


def test_softmax_rows_and_loss_backwards():
    g = _g(3)
    x = (torch.randn(6, 20, 1000, generator=g) * 4)
    x[x > 6] = O.MIN_VALUE
    got = ops.softmax_rows(x.cuda())
    np.testing.assert_allclose(got.cpu().numpy(), torch.softmax(x, -1).numpy(), rtol=2e-5, atol=1e-8)
    # masked log-softmax mean (utils.py:139,142): gradient against autograd
    lab = (torch.rand(6, 20, 1000, generator=g) < 0.002).float()
    xr = x.clone().requires_grad_(True)
    ref = -torch.masked_select(torch.log_softmax(xr, 2), lab.bool()).mean()
    (ref * 1.7).backward()
    xd = x.clone().cuda().requires_grad_(True)
    loss = ops.masked_lsm(xd, lab.cuda())
    (loss * 1.7).backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-7)
    # log_softmax(logits)[target] (utils.py:131-132)
    logits = torch.randn(37, 5000, generator=g) * 3
    tgt = torch.randint(0, 5000, (37,), generator=g)
    wts = torch.randn(37, generator=g)
    lr = logits.clone().requires_grad_(True)
    (torch.log_softmax(lr, 1).gather(1, tgt.unsqueeze(1)).squeeze(1) * wts).sum().backward()
    ld = logits.clone().cuda().requires_grad_(True)
    (ops.nll_gather(ld, tgt.cuda()) * wts.cuda()).sum().backward()
    np.testing.assert_allclose(ld.grad.cpu().numpy(), lr.grad.numpy(), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize('rows', [640, 77, 30720])
def test_batchnorm_train_relu_forward_backward(rows):
    """ops.bn_relu_train on [rows, C] against nn.BatchNorm1d (train mode) + ReLU autograd on the [B, C, Ft] layout the
    reference feeds it (model.py:397): output, running statistics, num_batches_tracked, dx, d weight, d bias."""
    g = _g(rows)
    Cc = 1024
    x = torch.randn(rows, Cc, generator=g) * 2 + 0.5
    bn = torch.nn.BatchNorm1d(Cc)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(Cc, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(Cc, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(Cc, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(Cc, generator=g) + 0.5)
    import copy
    bd = copy.deepcopy(bn).cuda().train()
    bn.train()
    w = torch.randn(rows, Cc, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = torch.relu(bn(xr.view(1, rows, Cc).permute(0, 2, 1)).permute(0, 2, 1).reshape(rows, Cc))
    (yr * w).sum().backward()
    xd = x.clone().cuda().requires_grad_(True)
    yd = ops.bn_relu_train(xd, bd)
    (yd * w.cuda()).sum().backward()
    np.testing.assert_allclose(yd.detach().cpu().numpy(), yr.detach().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(bd.running_mean.cpu().numpy(), bn.running_mean.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bd.running_var.cpu().numpy(), bn.running_var.numpy(), rtol=1e-5, atol=1e-6)
    assert int(bd.num_batches_tracked) == int(bn.num_batches_tracked) == 1
    scale = float(xr.grad.abs().max())
    assert float((xd.grad.cpu() - xr.grad).abs().max()) <= 3e-5 * scale + 1e-6
    for a, b in ((bd.weight.grad, bn.weight.grad), (bd.bias.grad, bn.bias.grad)):
        assert float((a.cpu() - b).abs().max()) <= 3e-5 * float(b.abs().max()) + 1e-5


def test_dropout_function_is_its_own_backward():
    torch.manual_seed(0)
    x = torch.randn(64, 20, 512).cuda().requires_grad_(True)
    y = ops.dropout(x, 0.5, True)
    keep = (y != 0)
    frac = float(keep.float().mean())
    assert 0.48 < frac < 0.52
    assert torch.equal(y[keep], (x.detach() * 2.0)[keep])
    w = torch.randn_like(y)
    (y * w).sum().backward()
    assert torch.equal(x.grad, torch.where(keep, w * 2.0, torch.zeros_like(w)))
    assert ops.dropout(x, 0.5, False) is x and ops.dropout(x, 0.0, True) is x
