"""-m gpu: every HIP kernel (called through the C-ABI wrappers) against the CPU oracle on seeded inputs.
Integer / index / boolean outputs must be bit-exact; fp32 outputs within the stated tolerance."""
import os

import numpy as np
import pytest
import torch

import gvd_amd
from gvd_amd import hip, ops
from oracle import gvd_oracle as O

pytestmark = pytest.mark.gpu


def _g(seed):
    return torch.Generator().manual_seed(seed)


def test_library_is_the_hip_one():
    assert b'gfx950' in hip.lib().gvd_version()
    assert hip.lib().gvd_abi_version() == hip.ABI_VERSION


def test_cpu_tensor_fails_loudly():
    with pytest.raises(hip.GvdHipError):
        ops.gemm_nt(torch.zeros(4, 32), torch.zeros(8, 32))


@pytest.mark.parametrize('M,N,K,act', [(4, 1024, 1024, 0), (32, 4096, 1536, 0), (256, 5000, 1024, 0),
                                       (70, 433, 2048, 1), (4000, 2048, 2048, 1), (640, 512, 1024, 0),
                                       # M <= 16: the weight-streaming skinny kernel (all MB / RPW variants)
                                       (1, 5000, 1024, 0), (8, 5000, 1024, 0), (16, 4096, 2048, 1), (13, 1001, 512, 1),
                                       (3, 1024, 1024, 0), (7, 512, 2048, 0)])
def test_gemm_nt(M, N, K, act):
    g = _g(M + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    ref = A.double() @ W.double().t() + b.double()
    if act:
        ref = ref.clamp(min=0)
    out = ops.gemm_nt(A.cuda(), W.cuda(), b.cuda(), act).cpu()
    # asymmetric random operands: a transposed / mis-tiled result cannot pass
    np.testing.assert_allclose(out.numpy(), ref.float().numpy(), rtol=1e-5, atol=2e-5)


def test_grounder_batched_masked():
    g = _g(5)
    B, M, R, K = 3, 20, 1000, 2048
    xt = torch.randn(B, M, K, generator=g) * 0.05
    feats = torch.relu(torch.randn(B, R, K, generator=g))
    mask = (torch.rand(B, M, R + 1, generator=g) < 0.3).to(torch.uint8)
    mb = torch.randn(B, M, generator=g)
    rb = torch.randn(B, M, R, generator=g)
    ref = O.grounder_dot(xt, feats, mask[:, :, 1:], mb.unsqueeze(2) + rb)
    out = ops.grounder_dot(xt.cuda(), feats.cuda(), mask.cuda()[:, :, 1:], mb.cuda(), rb.cuda()).cpu()
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-5, atol=1e-4)
    assert torch.equal(out == O.MIN_VALUE, mask[:, :, 1:].bool())
    # shared xt (visual words), 2-D mask broadcast over rows, per-class bias
    vis = torch.randn(433, K, generator=g) * 0.02
    pm = (torch.rand(B, R + 1, generator=g) < 0.2).to(torch.uint8)
    cb = torch.randn(433, generator=g)
    ref = O.grounder_dot(vis.unsqueeze(0).expand(B, 433, K).contiguous(), feats, pm[:, 1:], cb.view(1, -1, 1))
    out = ops.grounder_dot(vis.cuda(), feats.cuda(), pm.cuda()[:, 1:], cb.cuda(), None, xt_shared=True).cpu()
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('B', [1, 4, 7, 16, 17, 33, 64, 100, 128, 129, 200])
def test_lstm_cell(B):
    """nn.LSTMCell over in-place input blocks: the skinny kernel (<= 16 rows), the K-split 32 x 32-tile kernel (17 .. 128 rows:
    one / two / four row tiles, 128 .. 512 workgroups), the 64 x 64-tile kernel above."""
    g = _g(B)
    H, E = 1024, 512
    W = {'c.weight_ih': torch.randn(4 * H, E + H, generator=g) / 32, 'c.weight_hh': torch.randn(4 * H, H, generator=g) / 32,
         'c.bias_ih': torch.randn(4 * H, generator=g) * 0.1, 'c.bias_hh': torch.randn(4 * H, generator=g) * 0.1}
    fc, xt = torch.randn(B, H, generator=g), torch.randn(B, E, generator=g)
    h, c = torch.randn(B, H, generator=g) * 0.5, torch.randn(B, H, generator=g)
    rh, rc = O.lstm_cell(torch.cat([fc, xt], 1), h, c, W, 'c')
    d = {k: v.cuda() for k, v in W.items()}
    gates = torch.empty(B, 4 * H, device='cuda')
    oh, oc = ops.lstm_cell([fc.cuda(), xt.cuda()], [d['c.weight_ih'][:, :H], d['c.weight_ih'][:, H:]], h.cuda(),
                           d['c.weight_hh'], d['c.bias_ih'], d['c.bias_hh'], c.cuda(), gates_out=gates)
    # K = 2560-term fp32 dot products in a different summation order than oneDNN: few-ulp differences
    np.testing.assert_allclose(oh.cpu().numpy(), rh.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(oc.cpu().numpy(), rc.numpy(), rtol=1e-5, atol=1e-5)
    assert torch.isfinite(gates).all() and float(gates[:, :H].min()) >= 0.0      # i gate is a sigmoid


@pytest.mark.parametrize('B', [17, 40, 64, 97, 128])
def test_lstm_cell_k_split_kernel_with_the_decoder_operands(B):
    """The K-split cell kernel (csrc/gemm_ks.hip, 17 .. 128 rows) with the operands the token loops hand it: the loop-invariant
    fc gates as a per-row bias, outputs written into strided views of the BPTT's saved-state arrays, two / three K
    segments (a wave's quarter of K then starts inside a segment and crosses into the next) - against fp64 and against the
    64 x 64-tile kernel on the same rows inside a batch of 200 (another summation order: a few ulps)."""
    g = _g(1000 + B)
    H = 1024
    for ks in ((512,), (1024, 1024), (1024, 512), (128, 384)):
        K = sum(ks)
        w = torch.randn(4 * H, K + H, generator=g).cuda() / 32
        w_hh = w[:, K:]
        xs = [torch.randn(200, k, generator=g).cuda() for k in ks]
        ws, o = [], 0
        for k in ks:
            ws.append(w[:, o:o + k])
            o += k
        h, c = torch.randn(200, H, generator=g).cuda() * 0.5, torch.randn(200, H, generator=g).cuda()
        b_ih, b_hh = torch.randn(4 * H, generator=g).cuda() * 0.1, torch.randn(4 * H, generator=g).cuda() * 0.1
        rb = torch.randn(200, 4 * H, generator=g).cuda() * 0.3
        states = torch.zeros(3, 200, 2 * H, device='cuda')                       # outputs land in column blocks of a wider array
        gates = torch.empty(B, 4 * H, device='cuda')
        oh, oc = ops.lstm_cell([x[:B] for x in xs], ws, h[:B], w_hh, b_ih, b_hh, c[:B], rowbias=rb[:B], gates_out=gates,
                               h_out=states[1, :B, :H], c_out=states[1, :B, H:])
        bh, bc = ops.lstm_cell(xs, ws, h, w_hh, b_ih, b_hh, c, rowbias=rb)        # 200 rows: the 64 x 64-tile kernel
        G = (torch.cat(xs, 1)[:B].double() @ w[:, :K].double().t() + h[:B].double() @ w_hh.double().t() + b_ih.double() + b_hh.double()
             + rb[:B].double())
        i, f, gg, o_ = G.chunk(4, 1)
        rc = torch.sigmoid(f) * c[:B].double() + torch.sigmoid(i) * torch.tanh(gg)
        rh = torch.sigmoid(o_) * torch.tanh(rc)
        assert float((oh.double() - rh).abs().max()) < 1e-5 and float((oc.double() - rc).abs().max()) < 2e-5, ks
        assert float((oh - bh[:B]).abs().max()) < 1e-5 and float((oc - bc[:B]).abs().max()) < 2e-5
        assert torch.equal(states[1, :B, :H], oh) and not states[0].any() and not states[2].any() and not states[1, B:].any()
        assert float((gates[:, :H].double() - torch.sigmoid(i)).abs().max()) < 1e-5


@pytest.mark.parametrize('B,R,Ft', [(4, 1000, 10), (3, 1000, 480), (40, 1000, 10), (2, 37, 5), (300, 130, 3)])
def test_attention_step(B, R, Ft):
    """The streaming attention kernel against the oracle; (300, 130, 3) launches more than 192 MB -> the nontemporal
    instantiation."""
    g = _g(B * R + Ft)
    H, A = 1024, 512
    opt = gvd_amd.opts.default_opt(vocab_size=10)
    sd = gvd_amd.synth.init_state_dict(opt, seed=2, profile='trained_like')
    h = torch.randn(B, H, generator=g) * 0.5
    pool, p_pool = torch.randn(B, R, H, generator=g), torch.randn(B, R, A, generator=g)
    conv, p_conv = torch.randn(B, Ft, H, generator=g), torch.randn(B, Ft, A, generator=g)
    am = (torch.rand(B, R + 1, generator=g) < 0.2).to(torch.uint8)
    pm = (torch.rand(B, R + 1, generator=g) < 0.5).to(torch.uint8) | am
    if B > 2:
        am[1, 1:] = 1        # fully masked sample: uniform softmax over -1e8 logits (SURVEY §7 masks)
        pm[1, 1:] = 1
    r_att = O.attention_temporal(h, conv, p_conv, sd)
    r_att2, r_logits, r_q = O.attention_region(h, pool, p_pool, am[:, 1:], pm[:, 1:], sd)
    dv = {k: v.cuda() for k, v in sd.items() if k.startswith('core.attention')}
    q12 = ops.gemm_nt(h.cuda(), torch.cat([dv['core.attention.h2att.weight'], dv['core.attention2.h2att.weight']]),
                      torch.cat([dv['core.attention.h2att.bias'], dv['core.attention2.h2att.bias']]))
    np.testing.assert_allclose(q12[:, A:].cpu().numpy(), r_q.numpy(), rtol=1e-5, atol=1e-5)
    logits = torch.empty(B, R, device='cuda')
    amc, pmc = am.cuda(), pm.cuda()
    region = dict(feats=pool.cuda(), p_feats=p_pool.cuda(), q=q12[:, A:], w=dv['core.attention2.alpha_net.weight'].view(-1),
                  alpha_bias=dv['core.attention2.alpha_net.bias'], att_mask=amc[:, 1:], pnt_mask=pmc[:, 1:],
                  logits_out=logits)
    temporal = dict(feats=conv.cuda(), p_feats=p_conv.cuda(), q=q12[:, :A], w=dv['core.attention.alpha_net.weight'].view(-1),
                    alpha_bias=dv['core.attention.alpha_net.bias'])
    s, cr, ct = ops.attention_step(region, temporal, want_separate=True)
    np.testing.assert_allclose(ct.cpu().numpy(), r_att.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(cr.cpu().numpy(), r_att2.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(s.cpu().numpy(), (r_att + r_att2).numpy(), rtol=1e-4, atol=4e-5)
    np.testing.assert_allclose(logits.cpu().numpy(), r_logits.numpy(), rtol=1e-5, atol=2e-5)
    assert torch.equal(logits.cpu() == O.MIN_VALUE, pm[:, 1:].bool())


@pytest.mark.parametrize('M,N,K,act', [(4004, 512, 1024, 1), (4004, 448, 2048, 0), (1100, 1024, 512, 0)])
def test_gemm_few_tile_products_with_device_row_count(M, N, K, act):
    """The few-tile products of a small batch's compacted preamble (4000 region rows x N <= 512: fewer than 256 tiles of
    128 x 128) take the pipelined 64 x 64 kernel also when the row count lives on the device: live rows bitwise equal to the
    launch without a device count, rows past it untouched, values vs fp64."""
    g = _g(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    full = ops.gemm_nt(A, W, b, act)
    ref = A.double() @ W.double().t() + b.double()
    if act:
        ref = ref.clamp_min(0)
    assert float((full.double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    live = M - 333
    m_dev = torch.tensor([live], dtype=torch.int32, device='cuda')
    out = torch.full((M, N), 7.0, device='cuda')
    ops.gemm_nt(A, W, b, act, out=out, m_dev=m_dev)
    assert torch.equal(out[:live], full[:live])
    assert bool((out[(live + 63) // 64 * 64:] == 7.0).all())


def test_fused_side_kernels_of_the_inference_preamble():
    """gvd_fc_feature / gvd_loc_features / gvd_affine_relu_rows / gvd_zero_rows_outside_window against the ATen chains they
    replace (model.py:306-308, 357-360, 397, 303-305 + 401)."""
    import torch.nn.functional as F
    g = _g(123)
    B, Ft, D, S = 5, 7, 3072, 50
    segs = torch.randn(B, Ft, D, generator=g).cuda()
    num = torch.randint(0, 9, (B, 7), generator=g).cuda()
    W, b = (torch.randn(S, 4, generator=g) * 0.5).cuda(), torch.randn(S, generator=g).cuda()
    out = ops.fc_feature(segs, num, W, b, pad_to=32)
    ref = torch.cat([F.layer_norm(segs.mean(1), [D]), F.layer_norm(F.relu(F.linear(num[:, 3:7].float(), W, b)), [S])], -1)
    assert out.shape == (B, 3136) and float(out[:, D + S:].abs().max()) == 0.0
    np.testing.assert_allclose(out[:, :D + S].cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=2e-6)
    # location features through a row map with a device-side live count
    R, T = 30, 3
    ppls = (torch.rand(B, R, 7, generator=g) * 500).cuda()
    src = torch.randperm(B * R, generator=g)[:100].to(torch.int32).cuda()
    live = torch.tensor([77], dtype=torch.int32, device='cuda')
    loc = ops.loc_features(ppls, src, live, 100, T, ldo=32)
    # (the expectation is computed on the CPU, like the reference: ATen's GPU kernel turns `x / 720.` into a multiplication by
    # the rounded reciprocal, the CPU kernel - and gvd_loc_features - divide)
    pc = ppls.cpu().view(-1, 7)[src.cpu().long()]
    want = torch.cat([pc[:, :4] / 720., (pc[:, 4] * 1. / T).unsqueeze(-1)], 1)
    assert torch.equal(loc[:77, :5].cpu(), want[:77]) and float(loc[:77, 5:].abs().max()) == 0.0
    # BatchNorm1d(eval) + ReLU as an affine of the last axis
    bn = torch.nn.BatchNorm1d(1024).cuda().eval()
    with torch.no_grad():
        bn.weight.normal_(1, 0.2); bn.bias.normal_(0, 0.2); bn.running_mean.normal_(0, 0.5); bn.running_var.uniform_(0.5, 2)
        x = torch.randn(B, Ft, 1024, generator=g).cuda()
        want = torch.relu(bn(x.permute(0, 2, 1))).permute(0, 2, 1)
        scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        got = ops.affine_relu_rows_(x.clone(), scale.contiguous(), (bn.bias - bn.running_mean * scale).contiguous())
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=2e-6)
    # sampling window
    sidx = torch.tensor([[0, 7], [2, 5], [3, 3], [6, 7], [0, 1]]).cuda()
    y = torch.randn(B, Ft, 1024, generator=g).cuda()
    t = torch.arange(Ft, device='cuda').view(1, Ft)
    keep = (t >= sidx[:, 0:1]) & (t < sidx[:, 1:2])
    assert torch.equal(ops.zero_rows_outside_window_(y.clone(), sidx), y.masked_fill(~keep.unsqueeze(-1), 0))


def test_tanh_fast_error_bound():
    """The score kernels' tanh (gvd_common.h: tanh_fast, 11 issue slots instead of ocml's 31) stays within 2.5e-7
    ABSOLUTE of the real tanh over the whole line, saturates exactly and propagates NaN (AttModel.py:45, 90)."""
    g = _g(77)
    x = torch.cat([torch.randn(1 << 20, generator=g) * 1.5, torch.randn(1 << 18, generator=g) * 0.05,
                   torch.linspace(-20, 20, 400001), torch.tensor([0.0, -0.0, 1e-30, -1e-30, 88.0, -88.0, 1e30, -1e30,
                                                                  float('inf'), -float('inf')])])
    xd = x.cuda()
    yd = torch.empty_like(xd)
    hip.check(hip.lib().gvd_tanh_fast_f32(hip.ptr(xd), hip.ptr(yd), xd.numel(), hip.stream_ptr()), 'tanh_fast')
    y = yd.cpu().double()
    err = (y - torch.tanh(x.double())).abs().max().item()
    assert err <= 2.5e-7, err
    assert y[-1] == -1.0 and y[-2] == 1.0 and y[-3] == -1.0 and y[-4] == 1.0 and y[-5] == -1.0 and y[-6] == 1.0
    assert (y.abs() <= 1.0).all()
    nan = torch.full((4,), float('nan'), device='cuda')
    out = torch.empty_like(nan)
    hip.check(hip.lib().gvd_tanh_fast_f32(hip.ptr(nan), hip.ptr(out), 4, hip.stream_ptr()), 'tanh_fast')
    assert torch.isnan(out).all()


def test_top2_unk_rule_and_embed():
    g = _g(9)
    B, V, E, unk = 37, 5000, 512, 4999
    logits = torch.randn(B, V, generator=g) * 3
    logits[::3, unk] = 50.0                      # UNK wins in a third of the rows -> runner-up is taken
    logits[5, 17] = logits[5, 1234] = 60.0       # exact tie: lowest index wins
    emb = torch.randn(V, E, generator=g)
    lp = torch.log_softmax(logits, 1)
    v, i = lp.topk(2, 1)
    keep = i[:, 0] != unk
    it = torch.where(keep, i[:, 0], i[:, 1])
    it[5] = 17
    want_lp = lp.gather(1, it.view(-1, 1)).view(-1)
    seq = torch.zeros(B, 20, dtype=torch.int64, device='cuda')
    lps = torch.zeros(B, 20, device='cuda')
    xt = torch.empty(B, E, device='cuda')
    lg, em = logits.cuda(), emb.cuda()
    rc = hip.lib().gvd_logsoftmax_top2_embed(hip.ptr(lg), V, B, V, unk, hip.ptr(seq[:, 3:]), 20, hip.ptr(lps[:, 3:]), 20,
                                             hip.ptr(em), E, hip.ptr(xt), E, hip.stream_ptr())
    assert rc == 0
    assert torch.equal(seq[:, 3].cpu(), it)
    np.testing.assert_allclose(lps[:, 3].cpu().numpy(), want_lp.numpy(), atol=2e-6)
    assert torch.equal(xt.cpu(), torch.relu(emb[it]))
    assert int(seq[:, :3].abs().sum()) == 0 and int(seq[:, 4:].abs().sum()) == 0   # strided writes only
    # rows helper: lse / picked / top-k
    tgt = torch.randint(0, V, (B,), generator=g)
    lse, picked, tv, ti = ops.logsoftmax_rows(lg, tgt.cuda(), topk=5)
    np.testing.assert_allclose(lse.cpu().numpy(), torch.logsumexp(logits, 1).numpy(), rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(picked.cpu().numpy(), lp.gather(1, tgt.view(-1, 1)).view(-1).numpy(), atol=1e-5)
    rv, ri = lp.topk(5, 1)
    rows = [r for r in range(B) if r != 5]
    assert torch.equal(ti.cpu()[rows], ri[rows])
    np.testing.assert_allclose(tv.cpu().numpy(), rv.numpy(), atol=1e-5)


@pytest.mark.parametrize('B,seed', [(4, 1), (9, 2)])
def test_box_targets_bit_exact(B, seed):
    opt = gvd_amd.opts.default_opt(vocab_size=200)
    inp = gvd_amd.synth.trim_to_batch(gvd_amd.synth.make_inputs(opt, B, seed=seed, train=True))
    pm, fm = inp['pnt_mask'], inp['frm_mask']
    ref_ov = O.bbox_overlaps(inp['ppls'], inp['gt_boxes'], fm | pm[:, 1:].unsqueeze(-1))
    ref_st = O.sim_mat_target(ref_ov, inp['gt_boxes'][:, :, 5])
    ov, st = ops.iou_targets(inp['ppls'].cuda(), inp['gt_boxes'].cuda(), fm.cuda(), pm.cuda())
    assert torch.equal(ov.cpu(), ref_ov)            # same fp32 operation order -> bit-exact IoU
    assert torch.equal(st.cpu(), ref_st)
    Lc = 17
    roi, fms = ops.step_targets(ov, inp['mask_boxes'].cuda(), fm.cuda(), pm.cuda(), Lc)
    for t in range(Lc):
        assert torch.equal(roi[:, t].cpu(), O.roi_labels_for_step(inp['mask_boxes'][:, :, :, t + 1], ref_ov))
        assert torch.equal(fms[:, t].cpu(), O.frame_mask_for_step(inp['mask_boxes'][:, 0, :, t + 1], fm, pm))
    assert float(roi.sum()) > 0                     # positives exist (GT boxes are jittered proposals)


def test_masked_lsm_loss():
    g = _g(3)
    x = torch.randn(6, 20, 1000, generator=g) * 4
    x[x > 6] = O.MIN_VALUE
    lab = (torch.rand(6, 20, 1000, generator=g) < 0.002).float()
    ref = -torch.masked_select(torch.log_softmax(x, 2), lab.bool()).mean()
    out, _ = ops.masked_lsm_loss(x.cuda(), lab.cuda())
    assert abs(float(out) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))


# (5, 33, 1000, 10) and (5, 17, 2000, 10): more than 192 MiB of features per launch -> the NONTEMPORAL instantiation
# attn_partial_group_kernel<5, true> (attention.hip: `nt`), the one bench.py's beam section times at 64 segments x 2000 regions
@pytest.mark.parametrize('K,Bs,N,Ft', [(5, 3, 203, 10), (3, 2, 2000, 37), (2, 7, 100, 1), (4, 1, 64, 480),
                                       (5, 33, 1000, 10), (5, 17, 2000, 10), (3, 34, 1000, 10)])
def test_attention_beam_group_kernel_is_bitwise_the_row_kernel(K, Bs, N, Ft):
    """Beam search: the K beam rows of a sample share its features.  The grouped kernel (one workgroup per chunk and
    SAMPLE, features read once for K queries) must give bit-for-bit what the per-row kernel gives on the expanded rows."""
    g = _g(K * 100 + N)
    A, H = 512, 1024
    feats, p_feats = torch.randn(Bs, N, H, generator=g), torch.randn(Bs, N, A, generator=g)
    tf, tp = torch.randn(Bs, Ft, H, generator=g), torch.randn(Bs, Ft, A, generator=g)
    q = torch.randn(Bs * K, 2 * A, generator=g)
    w2, w1 = torch.randn(1, A, generator=g) * 0.1, torch.randn(1, A, generator=g) * 0.1
    b2, b1 = torch.randn(1, generator=g), torch.randn(1, generator=g)
    mask = (torch.rand(Bs * K, N + 1, generator=g) < 0.2).to(torch.uint8)
    mask[:, 0] = 0
    lo = torch.zeros(Bs * K, N).cuda()
    region = dict(feats=feats.cuda(), p_feats=p_feats.cuda(), q=q.cuda()[:, A:], w=w2.cuda(), alpha_bias=b2.cuda(),
                  att_mask=mask.cuda()[:, 1:], pnt_mask=mask.cuda()[:, 1:], logits_out=lo, group=K)
    temporal = dict(feats=tf.cuda(), p_feats=tp.cuda(), q=q.cuda()[:, :A], w=w1.cuda(), alpha_bias=b1.cuda(), group=K)
    out, cr, ct = ops.attention_step(region, temporal, want_separate=True)
    torch.cuda.synchronize()
    # the row kernel on explicitly expanded features (group = 0): the same bits
    lo2 = torch.zeros(Bs * K, N).cuda()
    region = dict(feats=feats.repeat_interleave(K, 0).cuda(), p_feats=p_feats.repeat_interleave(K, 0).cuda(),
                  q=q.cuda()[:, A:], w=w2.cuda(), alpha_bias=b2.cuda(), att_mask=mask.cuda()[:, 1:],
                  pnt_mask=mask.cuda()[:, 1:], logits_out=lo2)
    temporal = dict(feats=tf.repeat_interleave(K, 0).cuda(), p_feats=tp.repeat_interleave(K, 0).cuda(), q=q.cuda()[:, :A],
                    w=w1.cuda(), alpha_bias=b1.cuda())
    ref, rr, rt = ops.attention_step(region, temporal, want_separate=True)
    for a, b in ((out, ref), (cr, rr), (ct, rt), (lo, lo2)):
        assert torch.equal(a, b)


@pytest.mark.parametrize('barrier', ['counter', 'cg'])
@pytest.mark.parametrize('B,T', [(3, 10), (2, 480), (40, 37), (70, 12), (150, 6), (200, 9), (257, 5)])
def test_gru_persistent_kernel(B, T, barrier):
    """Persistent cooperative bi-GRU (2 layers) vs the oracle's explicit time-loop GRU; repeated launches must
    be bitwise repeatable (a cross-workgroup visibility race would show up as run-to-run differences)."""
    opt = gvd_amd.opts.default_opt(vocab_size=10)
    sd = gvd_amd.synth.init_state_dict(opt, seed=5)
    gru = torch.nn.GRU(1024, 512, 2, dropout=0.2, bidirectional=True, batch_first=True)
    gru.load_state_dict({k[len('context_enc.'):]: v for k, v in sd.items() if k.startswith('context_enc.')})
    x = torch.randn(B, T, 1024, generator=_g(B * T))
    with torch.no_grad():
        ref = O.gru_bidir_2layer_loop(x, sd)
        gru = gru.cuda().eval()
        xc = x.cuda()
        outs, flags = [], []
        for _ in range(3):
            outs.append(ops.gru_bidir_2layer(xc, gru, barrier=barrier))
            flags += ops.gru_bidir_2layer.last_sync
        torch.cuda.synchronize()
    for f in flags:                                   # counter barrier: no workgroup timed out
        assert not ops.sync_timed_out(f)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    np.testing.assert_allclose(outs[0].cpu().numpy(), ref.numpy(), rtol=1e-4, atol=5e-5)


def test_gru_result_does_not_depend_on_the_batch_tiling():
    """The three unit forms of the persistent bi-GRU (8 / 16 / 32 hidden units per workgroup, chosen by the batch size: up to
    64 rows / up to 128 / above) compute every output with the same k order and the same explicit fused multiply-adds: a
    sample's hidden sequence is bitwise the same whichever batch it travels in."""
    opt = gvd_amd.opts.default_opt(vocab_size=10)
    sd = gvd_amd.synth.init_state_dict(opt, seed=5)
    gru = torch.nn.GRU(1024, 512, 2, dropout=0.2, bidirectional=True, batch_first=True)
    gru.load_state_dict({k[len('context_enc.'):]: v for k, v in sd.items() if k.startswith('context_enc.')})
    gru = gru.cuda().eval()
    x = torch.randn(230, 9, 1024, generator=_g(77)).cuda()
    with torch.no_grad():
        full = ops.gru_bidir_2layer(x, gru)                                    # 8 batch tiles: 32-unit form
        parts = torch.cat([ops.gru_bidir_2layer(x[a:b].contiguous(), gru)     # 40 rows: 8-unit, 100: 16-unit, 90: 16-unit
                           for a, b in ((0, 40), (40, 140), (140, 230))])
    torch.cuda.synchronize()
    assert torch.equal(full, parts)


@pytest.mark.parametrize('B,T', [(3, 10), (2, 480), (40, 37), (70, 12), (200, 9), (257, 5)])
def test_lstm_persistent_kernel(B, T):
    """`--t_attn_mode bilstm` (model.py:145-149): the persistent bidirectional LSTM (2 layers, csrc/lstm_seq.hip) vs the
    oracle's explicit time-loop LSTM - one / two workgroup groups, 1 ... 8 batch tiles, a second 256-row slice; repeated
    launches bitwise repeatable; a sample's hidden sequence does not depend on the batch it travels in."""
    opt = gvd_amd.opts.default_opt(vocab_size=10, t_attn_mode='bilstm')
    sd = gvd_amd.synth.init_state_dict(opt, seed=5)
    lstm = torch.nn.LSTM(1024, 512, 2, dropout=0.2, bidirectional=True, batch_first=True)
    lstm.load_state_dict({k[len('context_enc.'):]: v for k, v in sd.items() if k.startswith('context_enc.')})
    x = torch.randn(B, T, 1024, generator=_g(B * T + 1))
    with torch.no_grad():
        ref = O.lstm_bidir_2layer_loop(x, sd)
        lstm = lstm.cuda().eval()
        xc = x.cuda()
        outs, flags = [], []
        for _ in range(3):
            outs.append(ops.lstm_bidir_2layer(xc, lstm, flags=flags))
        head = ops.lstm_bidir_2layer(xc[:min(B, 5)].contiguous(), lstm, flags=flags)
        torch.cuda.synchronize()
    assert all(int(f.sum()) == 0 for f in flags)          # no barrier timed out
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.equal(head, outs[0][:min(B, 5)])
    np.testing.assert_allclose(outs[0].cpu().numpy(), ref.numpy(), rtol=1e-4, atol=5e-5)


def test_lstm_layer_without_the_persistent_kernel(monkeypatch):
    """After a grid-barrier timeout the process runs without its persistent kernels (ops._persistent): the bilstm layer is then
    one fused LSTM-cell launch per (step, direction) - same outputs, saved gates and cell states as the persistent kernel
    within fp32 summation-order noise."""
    opt = gvd_amd.opts.default_opt(vocab_size=10, t_attn_mode='bilstm')
    sd = gvd_amd.synth.init_state_dict(opt, seed=5)
    g = lambda n: sd['context_enc.' + n].cuda().contiguous()
    B, T, Hh = 37, 9, 512
    gi = (torch.randn(B * T, 8 * Hh, generator=_g(9)) * 0.5).cuda()
    args = (gi, g('weight_hh_l0'), g('bias_hh_l0'), g('weight_hh_l0_reverse'), g('bias_hh_l0_reverse'), B, T, Hh)
    a = ops.lstm_seq_layer(*args, save=True)
    monkeypatch.setitem(ops._persistent, 'on', False)
    b = ops.lstm_seq_layer(*args, save=True)
    c = ops.lstm_seq_layer(*args)
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        np.testing.assert_allclose(y.cpu().numpy(), x.cpu().numpy(), rtol=1e-4, atol=2e-5)
    assert torch.equal(c, b[0])


@pytest.mark.parametrize('B,T', [(4, 10), (37, 6)])
def test_lstm_layer_autograd_matches_nn_lstm(B, T):
    """lstm_fn.lstm_bidir_2layer_train (persistent-kernel forward that keeps gates + cell states, hand-scheduled BPTT) against
    autograd through torch.nn.LSTM on the CPU in fp64: output, d input and every parameter gradient."""
    from gvd_amd import lstm_fn
    torch.manual_seed(3)
    ref = torch.nn.LSTM(1024, 512, 2, dropout=0.0, bidirectional=True, batch_first=True)
    mine = torch.nn.LSTM(1024, 512, 2, dropout=0.0, bidirectional=True, batch_first=True)
    mine.load_state_dict(ref.state_dict())
    ref, mine = ref.double(), mine.cuda()
    x = torch.randn(B, T, 1024, generator=_g(5))
    G = torch.randn(B, T, 1024, generator=_g(6))
    xr = x.double().requires_grad_(True)
    yr = ref(xr)[0]
    (yr * G.double()).sum().backward()
    xm = x.cuda().requires_grad_(True)
    ym = lstm_fn.lstm_bidir_2layer_train(xm, mine)
    (ym * G.cuda()).sum().backward()
    np.testing.assert_allclose(ym.detach().cpu().numpy(), yr.detach().numpy(), rtol=1e-4, atol=2e-5)
    rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-12))
    assert rel(xm.grad, xr.grad) < 1e-4
    for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        assert rel(p.grad, q.grad) < 1e-4, (n, rel(p.grad, q.grad))


def test_add_layernorm_unbiased():
    g = _g(11)
    x, y = torch.randn(777, 1024, generator=g), torch.randn(777, 1024, generator=g) * 0.3
    gamma, beta = 1 + 0.1 * torch.randn(1024, generator=g), 0.1 * torch.randn(1024, generator=g)
    ref = O.custom_layernorm(x + y, gamma, beta)
    out = ops.add_layernorm_unbiased(x.cuda(), y.cuda(), gamma.cuda(), beta.cuda(), 1e-6).cpu()
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    out = ops.add_layernorm_unbiased(x.cuda(), None, gamma.cuda(), beta.cuda(), 1e-6).cpu()
    np.testing.assert_allclose(out.numpy(), O.custom_layernorm(x, gamma, beta).numpy(), rtol=1e-5, atol=1e-5)


def test_region_feature_rows():
    """mask + class softmax + 3 layer norms + concat (model.py:336-364) in one kernel vs the op-by-op composition."""
    g = _g(12)
    B, R, D1 = 3, 203, 433
    g_pool = torch.relu(torch.randn(B, R, 2048, generator=g))
    loc = torch.relu(torch.randn(B, R, 300, generator=g))
    logits = torch.randn(B, D1, R, generator=g) * 3                 # reference layout [B,D1,R]
    pm = (torch.rand(B, R + 1, generator=g) < 0.3).to(torch.uint8)
    ml = logits.masked_fill(pm[:, 1:].bool().unsqueeze(1), O.MIN_VALUE)
    sim = torch.softmax(ml, dim=1)
    F = torch.nn.functional
    ref = torch.cat([F.layer_norm(g_pool, [2048]), F.layer_norm(loc, [300]),
                     F.layer_norm(sim.permute(0, 2, 1).contiguous(), [D1])], 2)
    out, sim_t = ops.region_feature_rows(g_pool.cuda(), loc.cuda(), logits.permute(0, 2, 1).contiguous().cuda(), pm.cuda())
    np.testing.assert_allclose(sim_t.cpu().numpy(), sim.permute(0, 2, 1).numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize('B,R', [(2, 1000), (3, 77), (1, 128), (2, 129), (1, 33)])
def test_flash_attention_padded_heads(B, R):
    """The padded-head attention kernel (flash_attn_pad.hip) behind the fused QKV projection: vs the reference's per-head
    bmm / softmax / bmm on the real 171 x 5 + 169 columns, pads of the output exactly zero."""
    g = _g(B * R + 5)
    D, nh, HP = 1024, 6, ops.HEAD_PAD
    q = torch.randn(B, R, D, generator=g) * 9.0
    k = torch.randn(B, R, D, generator=g)
    v = torch.randn(B, R, D, generator=g)
    sizes = [t.shape[-1] for t in q[:1, :1].chunk(nh, -1)]
    heads = []
    for qh, kh, vh in zip(q.chunk(nh, -1), k.chunk(nh, -1), v.chunk(nh, -1)):
        heads.append(torch.matmul(torch.softmax(torch.matmul(qh, kh.transpose(1, 2)) / 32.0, -1), vh))
    qkv = torch.zeros(B, R, 3 * nh * HP)
    c0 = 0
    for h, w in enumerate(sizes):
        for j, t in enumerate((q, k, v)):
            qkv[:, :, (j * nh + h) * HP:(j * nh + h) * HP + w] = t[:, :, c0:c0 + w]
        c0 += w
    o = ops.flash_attn_padded(qkv.cuda(), nh, 1.0 / 32.0).cpu()
    for h, w in enumerate(sizes):
        np.testing.assert_allclose(o[:, :, h * HP:h * HP + w].numpy(), heads[h].numpy(), rtol=2e-5, atol=2e-5)
        assert float(o[:, :, h * HP + w:(h + 1) * HP].abs().max()) == 0.0


def test_region_feature_rows_padded():
    """K-padded output rows (the pool_embed GEMM operand): same values, zero pad columns."""
    g = _g(13)
    B, R, D1 = 2, 131, 433
    g_pool = torch.relu(torch.randn(B, R, 2048, generator=g)).cuda()
    loc = torch.relu(torch.randn(B, R, 300, generator=g)).cuda()
    logits = (torch.randn(B, R, D1, generator=g) * 3).cuda()
    pm = (torch.rand(B, R + 1, generator=g) < 0.3).to(torch.uint8).cuda()
    a, sa = ops.region_feature_rows(g_pool, loc, logits, pm)
    b, sb = ops.region_feature_rows(g_pool, loc, logits, pm, pad_to=32)
    assert b.shape[-1] == 2784 and torch.equal(a, b[:, :, :2781]) and torch.equal(sa, sb)
    assert float(b[:, :, 2781:].abs().max()) == 0.0


def test_fused_encoder_path_matches_oracle():
    """obj_interact inference on the fused HIP path (one padded QKV GEMM, padded-head flash attention, own GEMMs, fused
    residual LayerNorm) vs the oracle's restatement of transformer.py:135-190 on the same weights - and again after an
    in-place weight change: the packed (re-laid-out) weight copies must follow."""
    from gvd_amd import att_model, synth
    opt = gvd_amd.opts.default_opt(vocab_size=300, t_attn_size=10)
    sd = synth.init_state_dict(opt, seed=21)
    m = att_model.TopDownModel(opt)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    x = torch.relu(torch.randn(3, 1000, 1024, generator=_g(3)))
    with torch.no_grad():
        ref = O.obj_interact(x, sd)
        got = m._obj_interact(x.cuda())
        # weights changed in place -> the packed copies must follow
        m.obj_interact.encoder.layers[0].selfattn.layer.wq.weight.mul_(0.5)
        sd2 = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        ref2 = O.obj_interact(x, sd2)
        got2 = m._obj_interact(x.cuda())
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(got2.cpu().numpy(), ref2.numpy(), rtol=1e-4, atol=1e-4)
    assert not torch.allclose(ref, ref2, atol=1e-3)


def _masked_inputs(B, R, D, seed, frac=0.25, full=(), none=()):
    g = _g(seed)
    pm = torch.zeros(B, R + 1, dtype=torch.uint8)
    pm[:, 1:] = (torch.rand(B, R, generator=g) < frac).to(torch.uint8)
    for b in full:
        pm[b, 1:] = 1
    for b in none:
        pm[b, 1:] = 0
    x = torch.randn(B, R, D, generator=g)
    x = x.masked_fill(pm[:, 1:].bool().unsqueeze(-1), 0.0)
    return pm, x


def test_compact_index_gather_expand():
    """gvd_compact_index / gvd_gather_rows_f32: per segment [valid rows in order | one representative masked row];
    a fully masked segment, a segment without masked rows; gather -> expand restores every row of a tensor whose
    masked rows are identical."""
    B, R, D = 5, 203, 36
    pm, x = _masked_inputs(B, R, D, 31, full=(1,), none=(3,))
    ci = ops.CompactIndex(pm.cuda())
    torch.cuda.synchronize()
    off, nv = ci.off.cpu().tolist(), ci.nvalid.cpu().tolist()
    src, cidx, repw, cm = ci.src_row.cpu(), ci.cidx.cpu().view(B, R), ci.rep_w.cpu(), ci.cmask.cpu()
    base = 0
    for b in range(B):
        valid = (pm[b, 1:] == 0).nonzero().view(-1)
        assert off[b] == base and nv[b] == valid.numel()
        assert src[base:base + valid.numel()].tolist() == (b * R + valid).tolist()
        assert cidx[b][valid].tolist() == list(range(base, base + valid.numel()))
        rep = base + valid.numel()
        nm = R - valid.numel()
        assert int(cm[rep]) == 1 and int(cm[base:rep].sum()) == 0
        if nm:
            first = int((pm[b, 1:] != 0).nonzero()[0])
            assert int(src[rep]) == b * R + first and abs(float(repw[b]) - np.log2(nm)) < 1e-6
            assert (cidx[b][pm[b, 1:] != 0] == rep).all()
        else:
            assert float(repw[b]) == float('-inf')
        base = rep + 1
    assert off[B] == base and int(ci.m_dev.cpu()) == base
    xc = ci.gather(x.cuda())
    back = ci.expand(xc)
    assert torch.equal(back.cpu(), x)
    # non-multiple-of-4 width (the 7 proposal columns)
    p7 = torch.randn(B, R, 7, generator=_g(1)).masked_fill(pm[:, 1:].bool().unsqueeze(-1), 0.0)
    assert torch.equal(ci.expand(ci.gather(p7.cuda())).cpu(), p7)
    # the precondition check
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    ops.check_masked_rows_zero(x.cuda(), pm.cuda(), flag)
    assert int(flag) == 0
    bad = x.clone()
    bad[1, 7, 3] = 1.0
    ops.check_masked_rows_zero(bad.cuda(), pm.cuda(), flag)
    assert int(flag) == 1


@pytest.mark.parametrize('M,N,K', [(33000, 1024, 1024), (40000, 433, 2048), (3000, 512, 1024)])
def test_gemm_device_side_row_count(M, N, K):
    """m_dev: tiles past the live row count are skipped (output untouched), live rows are bitwise the plain product."""
    g = _g(M)
    A = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    full = ops.gemm_nt(A, W, b, 1)
    for live in (M, M - 1, M * 4 // 5 + 3, 129, 1):
        out = torch.full((M, N), 7.0, device='cuda')
        ops.gemm_nt(A, W, b, 1, out=out, m_dev=torch.tensor([live], dtype=torch.int32, device='cuda'))
        assert torch.equal(out[:live], full[:live])
        assert bool((out[live:] == 7.0).all())


@pytest.mark.parametrize('B,R', [(3, 1000), (4, 130), (2, 77)])
def test_flash_attention_ragged_weighted_key(B, R):
    """Ragged padded-head attention over compacted rows == the dense kernel over rows with n identical masked rows (as
    keys: one key with weight n; as queries: one answer), incl. a fully masked and an unmasked segment."""
    nh, HP = 6, ops.HEAD_PAD
    pm, qkv = _masked_inputs(B, R, 3 * nh * HP, 7 * B + R, full=(1,), none=(0,))
    # masked rows must be IDENTICAL (not necessarily zero) for the equivalence: give them a common non-zero row
    common = torch.randn(3 * nh * HP, generator=_g(5)) * 0.5
    qkv = torch.where(pm[:, 1:].bool().unsqueeze(-1), common.view(1, 1, -1), qkv)
    qkv.view(B, R, 3 * nh, HP)[..., 171:] = 0
    dense = ops.flash_attn_padded(qkv.cuda(), nh, 1.0 / 32.0)
    ci = ops.CompactIndex(pm.cuda())
    comp = ops.flash_attn_padded(ci.gather(qkv.cuda()), nh, 1.0 / 32.0, ragged=(B, R + 1, ci.off, ci.rep_w))
    back = ci.expand(comp)
    np.testing.assert_allclose(back.cpu().numpy(), dense.cpu().numpy(), rtol=2e-5, atol=2e-6)


def test_compact_preamble_equals_dense_preamble():
    """TopDownModel._preamble on the compacted row set (GVD_COMPACT=1, default) vs the dense path: pool / p_pool /
    sim_mat, incl. a fully masked segment; masked rows of the dense tensors all carry their segment's representative."""
    import os
    from gvd_amd import att_model, synth
    from oracle import edge_cases
    opt, sd, inp = edge_cases.EDGE_CASES['masked_frames']()
    m = att_model.TopDownModel(opt)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    a = [inp[k].cuda() for k in ('segs_feat', 'num', 'ppls', 'ppls_feat', 'sample_idx', 'pnt_mask')]
    old = os.environ.get('GVD_COMPACT')
    try:
        with torch.no_grad():
            os.environ['GVD_COMPACT'] = '0'
            ref = m._preamble(*a, allow_compact=True)
            os.environ['GVD_COMPACT'] = '1'
            got = m._dense_regions(m._preamble(*a, allow_compact=True))
            m.check_kernel_status()
            # violated loader contract (a masked proposal with non-zero features) fails loudly
            bad = [t.clone() for t in a]
            bad[3][0, 2 * opt.num_prop_per_frm + 5, 11] = 1.0
            m._preamble(*bad, allow_compact=True)
            with pytest.raises(hip.GvdHipError):
                m.check_kernel_status()
    finally:
        if old is None:
            os.environ.pop('GVD_COMPACT', None)
        else:
            os.environ['GVD_COMPACT'] = old
    assert got['g_pool'] is None and ref['g_pool'] is not None
    for k in ('pool', 'p_pool', 'sim_mat_static', 'fc', 'conv', 'p_conv'):
        np.testing.assert_allclose(got[k].cpu().numpy(), ref[k].cpu().numpy(), rtol=1e-4, atol=2e-5, err_msg=k)
    pmb = inp['pnt_mask'][:, 1:].bool()
    for b in range(pmb.shape[0]):
        rows = got['pool'][b][pmb[b].cuda()]
        if rows.shape[0] > 1:
            assert torch.equal(rows, rows[:1].expand_as(rows))


@pytest.mark.parametrize('M,N,K', [(64000, 1024, 1024), (33024, 512, 1024), (64000, 1024, 2784), (40960, 2048, 2048)])
def test_backward_gemms_kstrided_operands(M, N, K):
    """dX = dY W and dW = dY^T X (nn.Linear backward) on the pipelined kernel with K-strided operands / batched split-K,
    vs fp64; and through autograd: ops.linear's gradients vs torch's."""
    g = _g(M + N + K)
    dY = torch.randn(M, N, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    X = torch.randn(M, K, generator=g).cuda()
    dx = ops.gemm_dx(dY, W)
    assert dx is not None
    rows = torch.cat([torch.arange(0, 200), torch.arange(M - 200, M)]).cuda()
    np.testing.assert_allclose(dx[rows].cpu().numpy(), (dY[rows].double() @ W.double()).float().cpu().numpy(), rtol=1e-5,
                               atol=1e-4)
    dw = ops.gemm_dw(dY, X)
    assert dw is not None
    want = (dY.double().t() @ X.double())
    err = float((dw.double() - want).abs().max() / want.abs().max())
    assert err < 1e-5, err      # fp32 accumulation over a 40k-64k long contraction
    # run-to-run bitwise reproducible (batched split-K + ordered sum, no atomics)
    assert torch.equal(dw, ops.gemm_dw(dY, X))


def test_linear_autograd_matches_torch():
    g = _g(9)
    M, N, K = 33024, 512, 1024
    x = torch.randn(M, K, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(N, K, generator=g) / 32).cuda().requires_grad_(True)
    b = torch.randn(N, generator=g).cuda().requires_grad_(True)
    up = torch.randn(M, N, generator=g).cuda()
    y = ops.linear(x, w, b, 1)
    y.backward(up)
    gx, gw, gb = x.grad.clone(), w.grad.clone(), b.grad.clone()
    x.grad = w.grad = b.grad = None
    torch.relu(torch.nn.functional.linear(x, w, b)).backward(up)
    np.testing.assert_allclose(gx.cpu().numpy(), x.grad.cpu().numpy(), rtol=1e-4, atol=1e-4)
    assert float((gw - w.grad).abs().max() / w.grad.abs().max()) < 1e-5
    np.testing.assert_allclose(gb.cpu().numpy(), b.grad.cpu().numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize('M,N,K', [(256, 5000, 1024), (256, 1024, 1024), (200, 4096, 3072), (96, 433, 2048), (512, 512, 1536)])
def test_gemm_small_pipelined_kernel(M, N, K):
    """Token-loop products (17..512 rows) run the pipelined 64 x 64 kernel (gemm_small.hip): vs fp64 and BITWISE vs the
    general kernel (32-row slices take its 32 x 128 tiles: same k order per output element)."""
    g = _g(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    out = ops.gemm_nt(A, W, b, 1)
    ref = (A.double() @ W.double().t() + b.double()).clamp(min=0)
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().cpu().numpy(), rtol=1e-5, atol=3e-5)
    for r0 in range(0, M - 16, 32):                     # (a last slice of <= 16 rows would take the skinny GEMV kernel)
        part = ops.gemm_nt(A[r0:r0 + 32].contiguous(), W, b, 1)
        assert torch.equal(part, out[r0:r0 + 32]), 'pipelined small-M kernel differs bitwise from the general kernel'


def test_cls_loss_fused_matches_torch():
    """T2: fused gather + log + masked mean of the region-classification loss (model.py:345-350), value and gradient."""
    g = _g(17)
    B, D1, R, K = 3, 433, 257, 5
    logits = (torch.randn(B, D1, R, generator=g) * 4).cuda().requires_grad_(True)
    tgt = torch.randint(0, D1, (B, K, R), generator=g)
    tgt[torch.rand(B, K, R, generator=g) < 0.7] = 0
    tgt = tgt.cuda()
    sim = torch.softmax(logits, 1)
    got = ops.cls_loss(sim, tgt)
    got.backward()
    g1 = logits.grad.clone()
    logits.grad = None
    sim2 = torch.softmax(logits, 1)
    p = torch.masked_select(torch.gather(sim2, 1, tgt), tgt > 0)
    want = -torch.clamp(torch.log(p), min=-100.0).mean()
    want.backward()
    assert abs(float(got) - float(want)) < 1e-5
    np.testing.assert_allclose(g1.cpu().numpy(), logits.grad.cpu().numpy(), rtol=1e-4, atol=1e-7)
    assert torch.equal(ops.cls_loss(sim.detach(), tgt), ops.cls_loss(sim.detach(), tgt))      # ordered reduction
    assert torch.isnan(ops.cls_loss(sim.detach(), torch.zeros_like(tgt)))                      # empty selection -> NaN


@pytest.mark.parametrize('rows', [1000, 130, 7])
def test_add_layernorm_fused_backward(rows):
    """Training ResidualBlock LayerNorm (transformer.py:66-88): the fused forward + backward row kernels against autograd
    through the module's elementwise formulation (fp64 reference for the gradients)."""
    from gvd_amd.att_model import _EncLayerNorm
    g = _g(23 + rows)
    D = 1024
    ln = _EncLayerNorm(D).cuda()
    with torch.no_grad():
        ln.gamma.copy_(torch.randn(D, generator=g) * 0.5 + 1)
        ln.beta.copy_(torch.randn(D, generator=g) * 0.1)
    x = torch.randn(rows, D, generator=g).cuda().requires_grad_(True)
    y = (torch.randn(rows, D, generator=g) * 0.3).cuda().requires_grad_(True)
    dout = torch.randn(rows, D, generator=g).cuda()
    out = ops.add_layernorm(x, y, ln.gamma, ln.beta, ln.eps)
    out.backward(dout)
    got = [t.grad.clone() for t in (x, y, ln.gamma, ln.beta)]
    for t in (x, y, ln.gamma, ln.beta):
        t.grad = None
    ln64 = _EncLayerNorm(D).cuda().double()
    with torch.no_grad():
        ln64.gamma.copy_(ln.gamma.double()); ln64.beta.copy_(ln.beta.double())
    x64, y64 = x.detach().double().requires_grad_(True), y.detach().double().requires_grad_(True)
    ref = ln64(x64 + y64)
    ref.backward(dout.double())
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().float().cpu().numpy(), rtol=2e-5, atol=2e-5)
    want = [x64.grad, y64.grad, ln64.gamma.grad, ln64.beta.grad]
    for a, b, name in zip(got, want, ('dx', 'dy', 'dgamma', 'dbeta')):
        scale = float(b.abs().max())
        err = float((a.double() - b).abs().max())
        assert err <= 2e-5 * max(scale, 1.0), (name, err, scale)
    assert torch.equal(got[0], got[1])


def _packed_qkv(B, Rp, nh, HP, g, d=1024, scale_q=1.0):
    """Random packed q | k | v [B, Rp, 3 nh HP] with the real 171 x 5 + 169 head widths (pad columns zero)."""
    sizes = [t.shape[-1] for t in torch.zeros(1, d).chunk(nh, -1)]
    qkv = torch.zeros(B, Rp, 3, nh, HP)
    for h in range(nh):
        qkv[:, :, :, h, :sizes[h]] = torch.randn(B, Rp, 3, sizes[h], generator=g)
    qkv[:, :, 0] *= scale_q
    return qkv.reshape(B, Rp, 3 * nh * HP), sizes


def _attn_core_reference(qkv, R, nh, HP, scale, key_bias=None, keep=None, p=0.0):
    """fp64 autograd formulation of transformer.py:90-117 per head over the packed layout: -> (O [B,R,nh*HP], leaf)."""
    B, Rp, _ = qkv.shape
    q64 = qkv.detach().double().view(B, Rp, 3, nh, HP).requires_grad_(True)
    outs = []
    for h in range(nh):
        qh, kh, vh = (q64[:, :R, j, h] for j in range(3))
        dots = torch.matmul(qh, kh.transpose(1, 2)) * scale
        if key_bias is not None:
            dots = dots + key_bias[:, :R].double().unsqueeze(1)
        w = torch.softmax(dots, -1)
        if keep is not None:
            w = w * keep[:, h, :R, :R].double() / (1 - p)
        outs.append(torch.matmul(w, vh))
    return torch.stack(outs, 2).reshape(B, R, nh * HP), q64


@pytest.mark.parametrize('B,R,p', [(2, 1000, 0.0), (3, 40, 0.0), (2, 1000, 0.2), (3, 132, 0.35)])
def test_enc_attn_core_training_matches_autograd(B, R, p):
    """ops.enc_attn_core - flash-style forward with in-register dropout, backward = one kernel that recomputes the
    probabilities from the saved logsumexp + three MFMA products - against the per-head fp64 autograd formulation of
    transformer.py:90-117 with the SAME keep mask (gvd_enc_dropout_mask re-evaluates the hash of csrc/enc_dropout.h):
    output and the gradient w.r.t. the packed q | k | v; pad rows zero; reproducible under the seed."""
    g = _g(B * R + int(100 * p))
    nh, HP = 6, ops.HEAD_PAD
    Rp = -(-R // 32) * 32
    qkv, _ = _packed_qkv(B, Rp, nh, HP, g, scale_q=3.0)
    qkv = qkv.cuda().requires_grad_(True)
    dO = torch.randn(B, Rp, nh * HP, generator=g).cuda()
    dO[:, R:] = 0                                        # pad rows never receive gradient
    seed = 0x1234_5678_9ABC_DEF0 + R
    out = ops.enc_attn_core(qkv, R, nh, 1.0 / 32, p, seed=seed)
    out.backward(dO)
    got = qkv.grad.clone()
    keep = ops.enc_dropout_mask(B * nh, Rp, p, seed).view(B, nh, Rp, Rp) if p > 0 else None
    ref, q64 = _attn_core_reference(qkv, R, nh, HP, 1.0 / 32, keep=keep, p=p)
    ref.backward(dO[:, :R].double())
    assert float((out[:, :R].double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    assert not out[:, R:].any()
    want = q64.grad.reshape(B, Rp, 3 * nh * HP)
    err = float((got.double() - want).abs().max())
    assert err < 5e-5 * float(want.abs().max()), err
    assert not got[:, R:].any()
    # same seed -> same bits; another seed -> another mask
    again = ops.enc_attn_core(qkv.detach(), R, nh, 1.0 / 32, p, seed=seed)
    assert torch.equal(again, out.detach())
    if p > 0:
        other = ops.enc_attn_core(qkv.detach(), R, nh, 1.0 / 32, p, seed=seed + 1)
        assert not torch.equal(other, out.detach())
        torch.manual_seed(5)
        o1 = ops.enc_attn_core(qkv.detach(), R, nh, 1.0 / 32, p)       # seed from torch's CPU generator
        torch.manual_seed(5)
        o2 = ops.enc_attn_core(qkv.detach(), R, nh, 1.0 / 32, p)
        assert torch.equal(o1, o2)


def test_enc_attn_core_with_and_without_the_score_map(monkeypatch):
    """The forward hands its scores to the backward maps kernel (ops.enc_core_scores) unless the map would exceed
    ops.ENC_SCORES_MAX_BYTES; then the backward multiplies Q K^T again.  Same output bits, gradients within the rounding of the
    two MFMA instructions that produce the scores (16x16x4 in the forward, 32x32x2 in the backward)."""
    g = _g(4242)
    nh, HP, B, R = 6, ops.HEAD_PAD, 2, 200
    Rp = -(-R // 32) * 32
    qkv, _ = _packed_qkv(B, Rp, nh, HP, g, scale_q=2.0)
    dO = torch.randn(B, Rp, nh * HP, generator=g).cuda()
    dO[:, R:] = 0
    res = []
    for cap in (ops.ENC_SCORES_MAX_BYTES, 0):
        monkeypatch.setattr(ops, 'ENC_SCORES_MAX_BYTES', cap)
        assert (ops.enc_core_scores(B, nh, Rp, 'cuda') is None) == (cap == 0)
        x = qkv.clone().cuda().requires_grad_(True)
        out = ops.enc_attn_core(x, R, nh, 1.0 / 32, 0.2, seed=77)
        out.backward(dO)
        res.append((out.detach().clone(), x.grad.clone()))
    assert torch.equal(res[0][0], res[1][0])
    assert float((res[0][1] - res[1][1]).abs().max()) < 2e-6 * float(res[0][1].abs().max())


@pytest.mark.parametrize('R,p,bias', [(1000, 0.2, True), (132, 0.0, False), (132, 0.35, True), (40, 0.2, False)])
def test_enc_attn_bwd_maps_against_fp64_and_inside_its_maps(R, p, bias):
    """gvd_enc_attn_bwd_maps alone (the epilogue stores straight from the accumulator layout through buffer descriptors that end with
    each (sample, head)'s map: blocks of a partial edge tile outside the map are dropped by the range check): both maps
    against the fp64 formulas with the SAME keep mask, exact zeros at rows / keys >= R, nothing written behind the maps
    (Rp = 160 and 64 are not multiples of the 128-wide tile)."""
    nh, HP, B = 6, ops.HEAD_PAD, 2
    Rp = -(-R // 32) * 32
    g = _g(R + int(100 * p))
    qkv, _ = _packed_qkv(B, Rp, nh, HP, g, scale_q=2.0)
    qkv = (qkv * 0.5).cuda()
    dO = torch.randn(B, Rp, nh * HP, generator=g).cuda()
    dO[:, R:] = 0
    Oo = torch.randn(B, Rp, nh * HP, generator=g).cuda()
    kb = (torch.randn(B, Rp, generator=g) * 0.5).cuda() if bias else None
    q64 = qkv.double().view(B, Rp, 3, nh, HP)
    S = torch.einsum('bqhd,bkhd->bhqk', q64[:, :R, 0], q64[:, :R, 1]) / 32
    if bias:
        S = S + kb[:, None, None, :R].double()
    lse = torch.logsumexp(S, -1)                                            # [B, nh, R] natural log
    lse2 = torch.zeros(B * nh, Rp, device='cuda')
    lse2.view(B, nh, Rp)[:, :, :R] = (lse * 1.4426950408889634).float()
    delta = torch.empty(B * nh, Rp, device='cuda')
    Pd = torch.full((B, nh, Rp, Rp), 7.0, device='cuda')
    dS = torch.full((B, nh, Rp, Rp), 7.0, device='cuda')
    guard = torch.full((1 << 18,), 3.0, device='cuda')
    seed = 991 + R
    # both forms: Q K^T multiplied again (scores = NULL), and the forward's log2-domain scores LOADED - rows >= R of that map are
    # never written by the forward: NaNs there must not reach the maps (the kernel's descriptor ends after row R - 1)
    sc = torch.full((B * nh, Rp, Rp), float('nan'), device='cuda')
    sc.view(B, nh, Rp, Rp)[:, :, :R, :R] = (S * 1.4426950408889634).float()
    sc.view(B, nh, Rp, Rp)[:, :, :R, R:] = 0.25                          # keys >= R: finite (the forward writes bias-only scores there)
    outs = []
    for scores in (None, sc):
        Pd.fill_(7.0), dS.fill_(7.0)
        hip.check(hip.lib().gvd_enc_attn_bwd_maps(hip.ptr(qkv), 3 * nh * HP, hip.ptr(dO), hip.ptr(Oo), nh * HP, hip.ptr(lse2), hip.ptr(kb) if bias else None,
                                                  hip.ptr(scores) if scores is not None else None,
                                                  hip.ptr(delta), hip.ptr(Pd), hip.ptr(dS), B, Rp, R, Rp, nh, HP, 1.0 / 32, p, seed, hip.stream_ptr()), 'maps')
        torch.cuda.synchronize()
        assert bool((guard == 3.0).all())
        for m in (Pd, dS):
            assert not m[:, :, R:].any() and not m[:, :, :, R:].any() and bool(torch.isfinite(m).all())
        outs.append((Pd.clone(), dS.clone()))
    assert float((outs[0][0] - outs[1][0]).abs().max()) < 2e-6 * float(outs[0][0].abs().max())
    assert float((outs[0][1] - outs[1][1]).abs().max()) < 2e-6 * float(outs[0][1].abs().max())
    keep = ops.enc_dropout_mask(B * nh, Rp, p, seed).view(B, nh, Rp, Rp)[:, :, :R, :R].double() if p > 0 else 1.0
    P = torch.exp(S - lse.unsqueeze(-1))
    dY = torch.einsum('bqhd,bkhd->bhqk', dO.double().view(B, Rp, nh, HP)[:, :R], q64[:, :R, 2])
    dl = (dO.double() * Oo.double()).view(B, Rp, nh, HP)[:, :R].sum(-1).permute(0, 2, 1)          # [B, nh, R]
    assert float((delta.view(B, nh, Rp)[:, :, :R].double() - dl).abs().max()) < 1e-4 * float(dl.abs().max())
    want_pd = P * keep / (1 - p)
    want_ds = P * (dY * keep / (1 - p) - dl.unsqueeze(-1)) / 32
    assert float((Pd[:, :, :R, :R].double() - want_pd).abs().max()) < 2e-5 * float(want_pd.abs().max())
    assert float((dS[:, :, :R, :R].double() - want_ds).abs().max()) < 5e-5 * float(want_ds.abs().max())


def test_enc_dropout_mask_statistics():
    """The keep mask of the attention dropout (counter-based hash, csrc/enc_dropout.h): keep rate, no structure along rows,
    columns or maps, different seeds independent."""
    n_maps, Rp, p = 12, 1024, 0.2
    m = ops.enc_dropout_mask(n_maps, Rp, p, 987654321).float()
    assert abs(float(m.mean()) - (1 - p)) < 5e-4                             # 12.6 M draws: sigma = 1.1e-4
    assert float((m.mean(dim=2) - (1 - p)).abs().max()) < 0.07               # every row (1024 draws: sigma = 0.0125)
    assert float((m.mean(dim=1) - (1 - p)).abs().max()) < 0.07               # every column
    c = m - m.mean()
    for a, b in ((c[:, :-1], c[:, 1:]), (c[:, :, :-1], c[:, :, 1:]), (c[:-1], c[1:]), (c[:, :-4, :-4], c[:, 4:, 4:])):
        assert abs(float((a * b).mean()) / float((c * c).mean())) < 2e-3      # neighbours uncorrelated
    m2 = ops.enc_dropout_mask(n_maps, Rp, p, 987654322).float()
    assert abs(float(((m2 - m2.mean()) * c).mean()) / float((c * c).mean())) < 2e-3
    assert float(ops.enc_dropout_mask(2, 64, 0.0, 1).float().mean()) == 1.0


def test_enc_dropout_rows_with_near_equal_additive_keys_do_not_share_a_shifted_mask():
    """with a 32-bit row key that enters the element hash by addition, two map rows whose keys differ by d < Rp carry
    the same mask shifted by d - at 64 x 6 maps of 1024 rows tens of thousands of such pairs exist.  The second key word (XORed
    onto the draw, csrc/enc_dropout.h) must bring their agreement down to chance: P(both kept) + P(both dropped) = 0.68 at
    p = 0.2.  The additive word is recomputed here on the host exactly as gvd_encdrop_row does."""
    n_maps, Rp, p, seed = 384, 1024, 0.2, 0x1234567899
    m = ops.enc_dropout_mask(n_maps, Rp, p, seed).view(-1, Rp).cpu().numpy()            # [rows, Rp] u8

    def mix32(x):
        x = x.astype(np.uint64)
        x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & np.uint64(0xffffffff)
        x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & np.uint64(0xffffffff)
        x ^= x >> np.uint64(16)
        return x
    lo, hi = np.uint64(seed & 0xffffffff), np.uint64(seed >> 32)
    rows = np.arange(n_maps * Rp, dtype=np.uint64)
    add = mix32(rows ^ hi) ^ lo
    order = np.argsort(add)
    d = np.diff(add[order].astype(np.int64))
    close = np.nonzero((d > 0) & (d < 256))[0]
    assert len(close) > 1000                         # the birthday pairs exist in numbers
    agree = []
    for i in close[:300]:
        a, b, dd = order[i], order[i + 1], int(d[i])           # add[b] = add[a] + dd: draw_b(key) shares mix32 with draw_a(key + dd)
        agree.append(float((m[b, :Rp - dd] == m[a, dd:]).mean()))
    assert abs(np.mean(agree) - 0.68) < 0.02, np.mean(agree)    # (identical shifted masks would give 1.0)


def test_enc_attn_core_key_bias():
    """The per-sample key bias of the training attention core (compacted training layout, train_compact.py): 0 leaves the
    result bit-identical to the unbiased call, log n weights a key n-fold, -inf removes it - forward and gradient against
    the fp64 formulation; heads of one sample share its bias row."""
    g = _g(77)
    B, R, nh, HP = 3, 96, 6, ops.HEAD_PAD
    Rp = R
    qkv, _ = _packed_qkv(B, Rp, nh, HP, g, scale_q=4.0)
    qkv = qkv.cuda().requires_grad_(True)
    dO = torch.randn(B, Rp, nh * HP, generator=g).cuda()
    plain = ops.enc_attn_core(qkv.detach(), R, nh, 1.0 / 32, 0.0)
    kb = torch.zeros(B, Rp)
    assert torch.equal(ops.enc_attn_core(qkv.detach(), R, nh, 1.0 / 32, 0.0, key_bias=kb.cuda()), plain)
    kb[0, 10] = float(np.log(7.0)); kb[0, 40:] = float('-inf')               # sample 0: key 10 counts 7-fold, keys 40.. absent
    kb[2, 0] = float(np.log(100.0)); kb[2, 1:5] = float('-inf')
    out = ops.enc_attn_core(qkv, R, nh, 1.0 / 32, 0.0, key_bias=kb.cuda())
    out.backward(dO)
    ref, q64 = _attn_core_reference(qkv, R, nh, HP, 1.0 / 32, key_bias=kb.cuda())
    ref.backward(dO.double())
    assert float((out.double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    want = q64.grad.reshape(B, Rp, 3 * nh * HP)
    assert float((qkv.grad.double() - want).abs().max()) < 5e-5 * float(want.abs().max())
    assert torch.equal(out[1], plain[1])                                     # sample 1 untouched
    kgrad = qkv.grad.view(B, Rp, 3, nh, HP)[:, :, 1:]                        # removed keys get no k / v gradient
    assert not kgrad[0, 40:].any() and not kgrad[2, 1:5].any()


@pytest.mark.parametrize('B,R', [(3, 40), (4, 40), (3, 42), (2, 111), (1, 2500)])
def test_encoder_training_path_matches_oracle_autograd(B, R):
    """The all-MFMA training encoder (one autograd function per layer: packed projection, flash-style attention core, fused
    LayerNorm forward + backward, residual gradients as dX addends) against the oracle's restatement of
    transformer.py:135-190 under autograd, eval mode: output, input gradient and every parameter gradient.  Shapes: (3, 40)
    B R % 32 != 0 -> region axis padded to 64; (4, 40) the rows of the batch packed back to back, no pad rows; (3, 42) and
    (2, 111) R % 4 != 0 -> padded layout with the rows R .. R4-1 masked as keys; (1, 2500) above 2048 padded rows (the
    16 KB key-bias stage of the flash-style core)."""
    from gvd_amd import att_model
    opt = gvd_amd.opts.default_opt(vocab_size=60)
    torch.manual_seed(3)
    model = att_model.TopDownModel(opt).cuda().eval()
    with torch.no_grad():                                  # LayerNorm parameters off their 1 / 0 initial values
        for lay in model.obj_interact.encoder.layers:
            for ln in (lay.selfattn.layernorm, lay.feedforward.layernorm):
                ln.gamma.add_(torch.randn(1024, device='cuda') * 0.1)
                ln.beta.add_(torch.randn(1024, device='cuda') * 0.1)
    g = _g(9)
    x = torch.randn(B, R, 1024, generator=g)
    dout = torch.randn(B, R, 1024, generator=g)
    names = [n for n, _ in model.named_parameters() if n.startswith('obj_interact.')]
    params = dict(model.named_parameters())
    xi = x.cuda().requires_grad_(True)
    with torch.enable_grad():
        out = model._obj_interact(xi)
        out.backward(dout.cuda())
    W = {n: params[n].detach().cpu().double().requires_grad_(True) for n in names}
    xr = x.double().requires_grad_(True)
    ref = O.obj_interact(xr, W)
    ref.backward(dout.double())
    assert float((out.detach().cpu().double() - ref.detach()).abs().max()) < 2e-4
    assert float((xi.grad.cpu().double() - xr.grad).abs().max()) < 2e-4 * max(1.0, float(xr.grad.abs().max()))
    for n in names:
        gb = W[n].grad
        assert float((params[n].grad.cpu().double() - gb).abs().max()) < 3e-4 * max(1.0, float(gb.abs().max())), n


@pytest.mark.parametrize('M,N,K,a_t', [(1000, 176, 1024, 0), (1000, 176, 1024, 1), (132, 176, 64, 1), (40, 192, 96, 0),
                                       (260, 132, 32, 1)])
def test_gemm_n192_kstrided_products(M, N, K, a_t):
    """The one-head-slot products of the training attention core's backward (csrc/gemm_n192.hip: W K-strided, 129..192 output
    columns, A plain or K-strided, two-level (sample, head) batch, edge rows / columns) against fp64."""
    g = _g(M + N + K)
    B, nh = 2, 3
    W = torch.randn(B, K, nh, N, generator=g).cuda()                                   # [K, N] per (b, h), row stride nh * N
    A = (torch.randn(B, nh, K, M, generator=g) if a_t else torch.randn(B, nh, M, K, generator=g)).cuda()
    out = torch.full((B, M + 3, nh, N), float('nan')).cuda()                           # rows past M / other heads: untouched
    ops._heads_bgemm(nh, A, 0, M if a_t else K, nh * M * K, M * K, W, 0, nh * N, K * nh * N, N, K, out, 0, nh * N,
                     (M + 3) * nh * N, N, M, N, B, a_t=a_t, w_t=1, what='n192')
    Ad = A.double().transpose(2, 3) if a_t else A.double()
    ref = torch.einsum('bhmk,bkhn->bmhn', Ad, W.double())
    err = float((out[:, :M].double() - ref).abs().max())
    assert err < 2e-5 * float(ref.abs().max()), err
    assert bool(torch.isnan(out[:, M:]).all())


def test_gemm_fused_row_gather_is_bitwise_the_gathered_gemm():
    """fc7 over the compacted proposal set: A rows read through a row map inside the pipelined GEMM vs gather + GEMM."""
    g = _g(77)
    src, M, N, K = 40000, 33000, 2048, 2048
    A = torch.randn(src, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) * 0.03).cuda()
    b = torch.randn(N, generator=g).cuda()
    rmap = torch.randint(0, src, (M,), generator=g).to(torch.int32).cuda()
    m_dev = torch.tensor([M - 700], dtype=torch.int32).cuda()
    want = ops.gemm_nt(A[rmap.long()].contiguous(), W, b, 1, m_dev=m_dev)
    got = ops.gemm_nt(A, W, b, 1, m_dev=m_dev, a_row_map=rmap)
    live = M - 700
    assert got.shape == (M, N) and torch.equal(got[:live], want[:live])
    with pytest.raises(hip.GvdHipError):                    # row offsets beyond the 32-bit buffer offset are refused
        big = torch.empty(600000, K, device='cuda')          # 4.9 GB of rows: offsets past 2^32
        ops.gemm_nt(big, W, b, 1, a_row_map=rmap)


@pytest.mark.parametrize('B', [8, 70])
def test_attention_row_map_reads_compacted_features(B):
    """The per-row attention kernel over COMPACTED region features (row map = CompactIndex.cidx) is bitwise the kernel
    over the dense expansion (masked rows are never fetched; a fully masked sample averages its representative)."""
    g = _g(B)
    R, H, A, Ft = 300, 1024, 512, 10
    pm = torch.zeros(B, R + 1, dtype=torch.uint8)
    pm[:, 1:] = (torch.rand(B, R, generator=g) < 0.3).to(torch.uint8)
    pm[1, 1:] = 1                                            # a fully masked sample
    pm[2, 1:] = 0                                            # a sample without masked rows
    pm = pm.cuda()
    ci = ops.CompactIndex(pm)
    live = int(ci.m_dev.item())
    pool_c = torch.zeros(ci.cap, H)
    pool_c[:live] = torch.randn(live, H, generator=g)
    p_pool_c = torch.zeros(ci.cap, A)
    p_pool_c[:live] = torch.randn(live, A, generator=g)
    pool_c, p_pool_c = pool_c.cuda(), p_pool_c.cuda()
    q = torch.randn(B, 2 * A, generator=g).cuda()
    w = (torch.randn(A, generator=g) * 0.2).cuda()
    ab = torch.zeros(1).cuda()
    conv, p_conv = torch.randn(B, Ft, H, generator=g).cuda(), torch.randn(B, Ft, A, generator=g).cuda()
    outs = []
    for compact in (False, True):
        lo = torch.empty(B, R, device='cuda')
        region = dict(feats=pool_c if compact else ci.expand(pool_c), p_feats=p_pool_c if compact else ci.expand(p_pool_c),
                      q=q[:, A:], w=w, alpha_bias=ab, att_mask=pm[:, 1:], pnt_mask=pm[:, 1:], logits_out=lo)
        if compact:
            region['row_map'] = ci.cidx
        temporal = dict(feats=conv, p_feats=p_conv, q=q[:, :A], w=w, alpha_bias=ab)
        outs.append((ops.attention_step(region, temporal), lo))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
