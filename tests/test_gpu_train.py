"""-m gpu: backward kernels against their torch stand-ins, the model's MLE gradients against the
reference's gradient norms (tests/golden), and the optimisation step."""
import os

import numpy as np
import pytest
import torch

import gvd_amd
from gvd_amd import att_model, ops, synth, train
from oracle import cases, edge_cases, gvd_oracle as O
from tests import torch_backend as TB

pytestmark = pytest.mark.gpu


def _g(seed):
    return torch.Generator().manual_seed(seed)


def test_lstm_cell_bwd_kernel():
    g = _g(1)
    B, H = 37, 1024
    gates = torch.cat([torch.rand(B, H, generator=g), torch.rand(B, H, generator=g),
                       torch.rand(B, H, generator=g) * 2 - 1, torch.rand(B, H, generator=g)], 1)
    dh, dc, cp, cn = (torch.randn(B, H, generator=g) for _ in range(4))
    rdg, rdc = TB.lstm_cell_bwd(dh, dc, gates, cp, cn)
    dg, dcp = ops.lstm_cell_bwd(dh.cuda(), dc.cuda(), gates.cuda(), cp.cuda(), cn.cuda())
    np.testing.assert_allclose(dg.cpu().numpy(), rdg.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dcp.cpu().numpy(), rdc.numpy(), rtol=1e-5, atol=1e-6)
    dg2, _ = ops.lstm_cell_bwd(dh.cuda(), None, gates.cuda(), cp.cuda(), cn.cuda())
    np.testing.assert_allclose(dg2.cpu().numpy(), TB.lstm_cell_bwd(dh, None, gates, cp, cn)[0].numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('B,N,masked', [(3, 1000, True), (2, 10, False), (33, 1000, True)])
def test_attn_bwd_kernels(B, N, masked):
    g = _g(B + N)
    H, A, Lc = 1024, 512, 3
    feats, p_feats = torch.randn(B, N, H, generator=g), torch.randn(B, N, A, generator=g)
    w = torch.randn(A, generator=g) * 0.2
    ab = torch.zeros(1)
    am = pm = None
    if masked:
        am = (torch.rand(B, N, generator=g) < 0.2).to(torch.uint8)
        pm = (torch.rand(B, N, generator=g) < 0.4).to(torch.uint8) | am
        am[0] = 1
        pm[0] = 1
    q_all = torch.randn(Lc, B, 2 * A, generator=g)
    de_all_ref, de_all = [], torch.empty(Lc, B, N, device='cuda')
    dev = lambda t: None if t is None else t.cuda()
    for t in range(Lc):
        q = q_all[t][:, A:]
        side = dict(feats=feats, p_feats=p_feats, q=q, w=w, alpha_bias=ab, att_mask=am, pnt_mask=pm)
        e = torch.tanh(p_feats + q.unsqueeze(1)) @ w
        if am is not None:
            e = e.masked_fill(am.bool(), -1e8)
        alpha = torch.softmax(e, 1)
        ctx = torch.bmm(alpha.unsqueeze(1), feats).squeeze(1)
        d_ctx, d_logits = torch.randn(B, H, generator=g), torch.randn(B, N, generator=g)
        r_de, r_dq, r_dw, r_dab = TB.attn_bwd_step(side, alpha, ctx, d_ctx, d_logits)
        dside = {k: dev(v) for k, v in side.items()}
        dside['q'] = q_all.cuda()[t][:, A:]
        de, dq, dw, dab = ops.attn_bwd_step(dside, alpha.cuda(), ctx.cuda(), d_ctx.cuda(), d_logits.cuda())
        np.testing.assert_allclose(de.cpu().numpy(), r_de.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(dq.cpu().numpy(), r_dq.numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(dw.cpu().numpy(), r_dw.numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(dab.cpu().numpy(), r_dab.numpy(), rtol=1e-4, atol=1e-4)
        de_all_ref.append(r_de)
        de_all[t].copy_(de)
    ref = TB.attn_bwd_pfeats(p_feats, q_all[:, :, A:], torch.stack(de_all_ref), w)
    out = ops.attn_bwd_pfeats(p_feats.cuda(), q_all.cuda()[:, :, A:], de_all, w.cuda())
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-5)


GRAD_CASES = [n for n, s in cases.CASES.items() if s['mode'] == 'MLE']

# Direction-aware gradient tolerances.  `projection_error` estimates |g - g_ref| / |g_ref| from 8 seeded random
# projections (cases.grad_projections); parameters whose true gradient is zero (the alpha_net biases: a softmax is shift
# invariant) hold only rounding noise on both sides and are exempt below NOISE x the largest gradient norm.
# Why 1e-3-level and not 1e-5: the forward activations of the two fp32 implementations differ by ~1e-6 relative
# (tools/fwd_diag.py: 1.5e-6 after pool_embed), so about one in a million ReLU pre-activations falls on the other side
# of zero; ONE such flip in the encoder's feed-forward (1.5 M pre-activations at B = 3) moves the gradient of that unit's
# bias by ~1 / sqrt(rows) and the Frobenius error of linear1.{weight,bias} to ~1e-3 - measured identically with every
# fused kernel switched to its library form (tools/grad_diag.py, profiles/r03/grad_diag_c.log), while the CPU oracle
# in fp32 stays within 1e-5 of its own fp64 run when no flip happens.  A sign-flipped / permuted / mis-routed gradient
# is off by O(1).
PROJ_TOL = 5e-3
ELEM_TOL = 3e-3
NOISE = 1e-6


def _check_projections(named_grads, names, norms, projs, what='grad', f64=None):
    """f64 = (norms, projections) of the fp64 value of the same gradients (cases with `f64_grads`): a parameter that misses the
    fp32 reference must then (i) agree with the fp64 value within the same bound and (ii) be explained by the reference's OWN
    distance from fp64 (its fp32 rounding - a ReLU pre-activation on the other side of zero - not a routing error); the
    exceptions are returned so that the caller can assert which ones it expects."""
    gmax = float(max(norms))
    worst = (0.0, None)
    excused = {}
    for i, (n, want_norm, want_proj) in enumerate(zip(names, norms, projs)):
        g = named_grads[n]
        assert g is not None, n
        if want_norm <= NOISE * gmax:
            continue
        err = cases.projection_error(n, g, want_proj, want_norm)
        if err >= PROJ_TOL and f64 is not None:
            err64 = cases.projection_error(n, g, f64[1][i], f64[0][i])
            ref_vs_64 = float(np.sqrt(((np.asarray(f64[1][i]) - np.asarray(want_proj)) ** 2).mean())) / float(want_norm)
            assert err64 < PROJ_TOL, '%s: %s off the fp32 reference by %.3g AND off the fp64 value by %.3g' % (n, what, err, err64)
            assert ref_vs_64 > 0.5 * err, '%s: %s off the fp32 reference by %.3g, which itself is only %.3g from fp64' % (n, what, err, ref_vs_64)
            excused[n] = (err, err64, ref_vs_64)
            continue
        worst = max(worst, (err, n))
        assert err < PROJ_TOL, '%s: %s projections off by %.3g x |reference| (direction / routing error?)' % (n, what, err)
    print('worst %s projection error %.3g (%s)' % (what, worst[0], worst[1]))
    for n, e in excused.items():
        print('%s: %.3g from the fp32 reference, %.3g from fp64; the fp32 reference itself is %.3g from fp64' % ((n,) + e))
    return excused


@pytest.mark.parametrize('name', GRAD_CASES)
def test_mle_gradients_match_reference(name, golden_dir):
    """Gradients of lm + w_att2*att2 + w_grd*grd + w_cls*cls against the REFERENCE's own backward (oracle/make_golden.py)
    at B = 4 ... 64: the 4 losses, every parameter's gradient L2 norm AND its direction (seeded random projections of the
    reference gradient, so a sign-flipped / permuted / mis-routed gradient of the right magnitude fails).  Eval-mode
    arithmetic, except the `bn_train` case: train mode with every dropout ratio 0, i.e. BatchNorm batch statistics + the
    running-statistics update (model.py:114,397)."""
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    spec = cases.CASES[name]
    opt, sd, inp = cases.build_case(name)
    assert cases.weight_fingerprint(sd) == int(g['weight_fp']) and cases.input_fingerprint(inp) == int(g['input_fp'])
    model = att_model.TopDownModel(opt)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    if spec.get('bn_train'):
        cases.zero_dropout(model).train()
    lm, a2, gl, cl = model(*synth.as_args(inp, 'cuda'), 'MLE')
    np.testing.assert_allclose(np.array([float(lm), float(a2), float(gl), float(cl)]), g['losses'], atol=1e-4)
    if spec.get('bn_train'):
        bn = model.att_embed_aux[0]
        np.testing.assert_allclose(bn.running_mean.cpu().numpy(), g['bn_running_mean'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(bn.running_var.cpu().numpy(), g['bn_running_var'], rtol=1e-4, atol=1e-5)
    w = cases.GRAD_WEIGHTS
    (lm.sum() + w['w_att2'] * a2.sum() + w['w_grd'] * gl.sum() + w['w_cls'] * cl.sum()).backward()
    model.check_kernel_status()
    names = [str(n) for n in g['grad_names']]
    ref = dict(zip(names, g['grad_norms']))
    params = dict(model.named_parameters())
    worst = 0.0
    for n, want in ref.items():
        p = params[n]
        assert p.grad is not None, n
        got = float(p.grad.double().norm())
        # absolute floor: alpha_net.bias of a softmax attention has an exactly-zero true gradient (shift
        # invariance); both sides only hold ~1e-8 rounding noise there
        rel = abs(got - want) / max(want, 1e-3)
        worst = max(worst, rel)
        assert rel < 2e-3, '%s: |grad| %.6g vs reference %.6g' % (n, got, want)
    f64 = (g['grad_norms_f64'], g['grad_proj_f64']) if spec.get('f64_grads') else None
    excused = _check_projections({n: params[n].grad for n in names}, names, g['grad_norms'], g['grad_proj'], f64=f64)
    # (seed 24 of the transfer_mode='none' case: the reference's fp32 gradient of the fc7 layer is 0.93 % from its fp64 value;
    # nothing else may need the fp64 excuse)
    assert set(excused) <= {'ctx2pool_grd.0.weight', 'ctx2pool_grd.0.bias'}, excused
    for n in ('core.i2h_2.weight', 'core.h2h_2.weight'):
        assert params[n].grad is None or float(params[n].grad.abs().sum()) == 0.0
    if 'att_input_mode' in spec.get('opt', {}):
        # parameters the reference's backward leaves without a gradient under this mode (the frame-wise encoder + attention
        # under 'region'): None or exactly zero here
        for n, p in params.items():
            if n not in ref:
                assert p.grad is None or float(p.grad.abs().sum()) == 0.0, n
    print('worst relative grad-norm error', worst)


def test_train_compaction_matches_full_rows(golden_dir, monkeypatch):
    """GVD_TRAIN_COMPACT=1 (train_compact.py: per segment its valid proposals + one weighted representative of the masked
    ones, Rc = 832 of R = 1000 rows here) against the full row set on the reference case mle_b4_v1000_ft10_trained: the
    four losses (also vs the reference's own) and every parameter gradient.  The maths is pinned in fp64 on the CPU
    (tests/test_train_compact_cpu.py); this is the HIP side: key-bias operand of the encoder's softmax row kernel,
    compacted targets / masks / token loop."""
    name = 'mle_b4_v1000_ft10_trained'
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    opt, sd, inp = cases.build_case(name)
    args = synth.as_args(inp, 'cuda')
    w = cases.GRAD_WEIGHTS
    res = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('GVD_TRAIN_COMPACT', mode)
        model = att_model.TopDownModel(opt)
        model.load_state_dict(sd)
        model = model.cuda().eval()
        lm, a2, gl, cl = model(*args, 'MLE')
        (lm.sum() + w['w_att2'] * a2.sum() + w['w_grd'] * gl.sum() + w['w_cls'] * cl.sum()).backward()
        model.check_kernel_status()
        res[mode] = (np.array([float(lm.detach()), float(a2.detach()), float(gl.detach()), float(cl.detach())]),
                     {n: p.grad.detach().double() for n, p in model.named_parameters() if p.grad is not None})
        np.testing.assert_allclose(res[mode][0], g['losses'], atol=1e-4)
    np.testing.assert_allclose(res['1'][0], res['0'][0], rtol=0, atol=2e-6)
    assert set(res['0'][1]) == set(res['1'][1])
    gmax = max(float(v.norm()) for v in res['0'][1].values())
    for n, a in res['0'][1].items():
        if float(a.norm()) > 1e-6 * gmax:
            err = float((a - res['1'][1][n]).norm() / a.norm())
            assert err < PROJ_TOL, (n, err)      # measured: 1.1e-5 worst (ctx2att.bias); one ReLU boundary flip costs ~1e-3
    from gvd_amd import train_compact
    c = train_compact.compact_regions(args[4], args[7], args[10], args[8])
    assert c is not None and c['Rc'] < args[4].shape[1]       # the case really runs compacted


@pytest.mark.parametrize('name', sorted(edge_cases.TRAIN_EDGE_CASES))
def test_mle_edge_shapes_match_oracle(name):
    """Training edge shapes (oracle/edge_cases.py: one segment, sizes no tile divides, an annotated frame with every
    proposal masked): 4 losses within 1e-4 of the oracle (itself pinned to the reference on these cases by
    tests/test_oracle_vs_reference.py) and EVERY parameter gradient elementwise against the oracle's autograd:
    relative Frobenius error <= ELEM_TOL and cosine >= 1 - ELEM_TOL^2 / 2 (zero-true-gradient parameters exempt)."""
    opt, sd, inp = edge_cases.TRAIN_EDGE_CASES[name]()
    W = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v)
         for k, v in sd.items()}
    w = cases.GRAD_WEIGHTS
    olm, oa2, ogl, ocl, _ = O.forward_train(W, opt, *[inp[k] for k in synth.FORWARD_ORDER])
    (olm + w['w_att2'] * oa2 + w['w_grd'] * ogl + w['w_cls'] * ocl).backward()
    model = att_model.TopDownModel(opt)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    lm, a2, gl, cl = model(*synth.as_args(inp, 'cuda'), 'MLE')
    np.testing.assert_allclose(np.array([float(lm), float(a2), float(gl), float(cl)]),
                               np.array([olm.item(), oa2.item(), ogl.item(), ocl.item()]), atol=1e-4)
    (lm.sum() + w['w_att2'] * a2.sum() + w['w_grd'] * gl.sum() + w['w_cls'] * cl.sum()).backward()
    gmax = max(float(v.grad.double().norm()) for v in W.values() if torch.is_tensor(v) and v.grad is not None)
    worst = (0.0, None)
    for n, p in model.named_parameters():
        want = None if W[n].grad is None else W[n].grad.double()
        got = None if p.grad is None else p.grad.double().cpu()
        wn = 0.0 if want is None else float(want.norm())
        gn = 0.0 if got is None else float(got.norm())
        assert abs(gn - wn) / max(wn, 1e-3) < 2e-3, '%s: |grad| %.6g vs oracle %.6g' % (n, gn, wn)
        if wn <= NOISE * gmax:        # unused (i2h_2 / h2h_2) or zero-true-gradient parameters: nothing to compare
            continue
        rel = float((got - want).norm()) / wn
        cos = float((got * want).sum()) / (wn * gn)
        worst = max(worst, (rel, n))
        assert rel <= ELEM_TOL and cos >= 1.0 - 0.5 * ELEM_TOL ** 2, \
            '%s: gradient differs elementwise: rel %.3g, cos %.9f' % (n, rel, cos)
    print('worst elementwise relative gradient error %.3g (%s)' % worst)


STEP_CASES = [n for n, s in cases.CASES.items() if s['mode'] == 'step']


@pytest.mark.parametrize('name', STEP_CASES)
def test_one_optimisation_step_matches_reference(name, golden_dir):
    """main.train's step (main.py:234-266 with the optimizer of 660-677) through train.Trainer on the HIP path vs the
    reference's own step (tests/golden/step_*.npz, eval-mode arithmetic): loss assembly, clip_grad_norm_(0.1) (the
    pre-clip total norm), Adam with lr x0.1 for the fc7 / vis_embed groups -> every parameter's first-moment norm
    (linear in the clipped gradient) and update norm."""
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    opt, sd, inp = cases.build_case(name)
    for k, v in cases.GRAD_WEIGHTS.items():
        setattr(opt, k, v)
    model = att_model.TopDownModel(opt)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    tr = train.Trainer(model, opt)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    losses = tr.step(synth.as_args(inp, 'cuda')).cpu().numpy()
    np.testing.assert_allclose(losses, g['losses'], atol=1e-4)
    assert abs(tr.last_grad_norm - float(g['total_grad_norm'])) / float(g['total_grad_norm']) < 2e-3
    lrs = {id(p): gr['lr'] for gr in tr.optimizer.param_groups for p in gr['params']}
    params = dict(model.named_parameters())
    assert abs(lrs[id(params['ctx2pool_grd.0.weight'])] - 5e-5) < 1e-12 and abs(lrs[id(params['logit.weight'])] - 5e-4) < 1e-12
    for n, dn, mn in zip([str(x) for x in g['step_names']], g['delta_norms'], g['exp_avg_norms']):
        p = params[n]
        got_m = float(tr.optimizer.state[p]['exp_avg'].double().norm())
        got_d = float((p.detach() - before[n]).double().norm())
        assert abs(got_m - mn) / max(mn, 1e-7) < 3e-3, '%s: |exp_avg| %.6g vs reference %.6g' % (n, got_m, mn)
        # the first Adam update is lr * g / (|g| + eps): elements whose gradient is rounding noise (~1e-8, e.g. the bias
        # of a softmax-attention alpha_net) can take either sign, so the update norm gets a looser, lr-scaled bound
        if mn > 1e-6:      # (the softmax-shift-invariant alpha_net biases have an exactly-zero true gradient)
            assert abs(got_d - dn) <= 0.02 * dn + 1e-9, '%s: |delta| %.6g vs reference %.6g' % (n, got_d, dn)
    # direction of the first moments (linear in the clipped gradient) and of the parameter updates themselves
    names = [str(x) for x in g['step_names']]
    _check_projections({n: tr.optimizer.state[params[n]]['exp_avg'] for n in names}, names, g['exp_avg_norms'],
                       g['exp_avg_proj'], what='exp_avg')
    big = [i for i, mn in enumerate(g['exp_avg_norms']) if mn > 1e-6]
    for i in big:
        n = names[i]
        err = cases.projection_error(n, params[n].detach() - before[n], g['delta_proj'][i], g['delta_norms'][i])
        # (elements whose gradient is rounding noise move by +-lr either way: looser than the first-moment bound)
        assert err < 0.05, '%s: update direction off by %.3g x |reference update|' % (n, err)
    for n in ('core.i2h_2.weight', 'core.h2h_2.weight'):
        assert torch.equal(params[n].detach(), before[n])


TRAJ_CASES = [n for n, s in cases.CASES.items() if s['mode'] == 'traj']
# bounds of the optimiser-state pins after step 4 (measured values: see the test's printout in profiles/r06/)
# = 1.5 x the largest value measured over the 81 parameters (session C: exp_avg projections 0.059, norm 0.022; exp_avg_sq
# projections 0.074, norm 0.048 - the same Adam amplification of rounding-level gradient differences that the losses show:
# tools/adam_noise_amplification.py reproduces these sizes between two CPU oracle runs, profiles/r06/adam_noise_amplification.txt)
M_NORM_TOL, M_PROJ_TOL = 3.3e-2, 9e-2
V_NORM_TOL, V_PROJ_TOL = 7.3e-2, 1.1e-1
STEP_DN_TOL = 2e-2


@pytest.mark.parametrize('name', TRAJ_CASES)
def test_optimisation_trajectory_matches_reference(name, golden_dir):
    """FOUR consecutive steps of main.train (main.py:234-266, 660-677) through train.Trainer against the reference's own
    four steps (tests/golden/traj4_*.npz, eval-mode arithmetic, a different batch per step): steps 2..4 run on the
    parameters the HIP path's OWN earlier updates produced, so the losses and the pre-clip gradient norm of every step pin
    the optimiser state across steps, not one update.  Bounds: the first step sees identical weights (1e-4, BASELINE).  Later
    steps carry the fp32 noise of the earlier updates THROUGH ADAM: in its first steps the update is lr * m / sqrt(v) = +-lr
    for every element whatever the size of its gradient, so an element whose gradient sits at the rounding-noise floor - where
    two fp32 implementations disagree on the sign - moves by lr in opposite directions on the two sides, step after step.  The
    LM loss differs by up to 4.8e-4 (step 3), the attention / grounding losses (softmaxes over nearly flat region logits)
    by up to 6.1e-4 (step 4 ; measured, round 4 sessions I - K; the CPU oracle, whose arithmetic order is the
    reference's, stays within 1e-4: tests/test_oracle_golden.py).  Asserted: 1e-4 at step 1, 2e-3 afterwards, i.e. < 0.5 % of
    the loss change a step makes.  The pre-clip gradient norm is a derivative and moves more than the losses: 1.03 % at
    step 4 on the device (10.264 vs 10.159, session J).  That this is the amplification and not an optimiser-state error is
    shown on the CPU alone: tools/adam_noise_amplification.py runs the ORACLE's four steps twice, exactly and with 1e-3
    relative Gaussian noise on every gradient (the measured HIP-vs-reference direction error), and the two runs differ by
    5e-4 in the losses and 0.82 % in the gradient norm at step 4 (profiles/r04/adam_noise_amplification.txt) - the same sizes.
    Gradient norm asserted: 2e-3 at step 1 (identical weights), 3e-2 afterwards.  Direction of the accumulated parameter change over the four steps: projection error below
    0.1 of the reference change for every parameter with a real gradient."""
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    opt, sd, _ = cases.build_case(name)
    for k, v in cases.GRAD_WEIGHTS.items():
        setattr(opt, k, v)
    model = att_model.TopDownModel(opt)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    tr = train.Trainer(model, opt)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    worst = 0.0
    step_dn = []
    for i, batch in enumerate(cases.traj_batches(name)):
        prev = [p.detach().clone() for p in model.parameters()]
        losses = tr.step(synth.as_args(batch, 'cuda')).cpu().numpy()
        step_dn.append(float(torch.sqrt(sum(((p.detach() - q).double() ** 2).sum() for p, q in zip(model.parameters(), prev)))))
        d = float(np.abs(losses - g['step_losses'][i]).max())
        worst = max(worst, d)
        print('step %d: |loss - reference| %s' % (i + 1, ['%.2e' % x for x in np.abs(losses - g['step_losses'][i])]))
        assert d < (1e-4 if i == 0 else 2e-3), 'step %d: losses %s vs reference %s' % (i, losses, g['step_losses'][i])
        want = float(g['step_grad_norms'][i])
        assert abs(tr.last_grad_norm - want) / want < (2e-3 if i == 0 else 3e-2), 'step %d: |grad| %.6g vs %.6g' % (i, tr.last_grad_norm, want)
    print('largest loss difference over the %d steps: %.3g' % (len(g['step_losses']), worst))
    params = dict(model.named_parameters())
    names = [str(x) for x in g['step_names']]
    big = float(max(g['delta_norms']))
    for i, n in enumerate(names):
        dn = float(g['delta_norms'][i])
        if dn < 1e-3 * big:
            continue                      # (parameters whose gradient is rounding noise: Adam moves them by +-lr either way)
        err = cases.projection_error(n, params[n].detach() - before[n], g['delta_proj'][i], dn)
        assert err < 0.1, '%s: accumulated update off by %.3g x |reference update|' % (n, err)
    # ---- the optimiser STATE after step 4, against the reference's own torch.optim.Adam (main.py:660-677): the step count of
    # every parameter (exact), the first moment (linear in the four clipped gradients: carries their direction error, ~1e-3
    # at step 1, and the drift of steps 2..4) and the second moment (quadratic: dominated by the large entries, the most
    # robust of the three) as norm + seeded projections; and the norm of EVERY step's whole update (the bias-corrected step
    # size lr * m_hat / (sqrt(v_hat) + eps) of steps 1..4 - a wrong bias-correction count at step t changes it by
    # (1 - b1^t') / (1 - b1^t) * sqrt((1 - b2^t) / (1 - b2^t')): 23 % between t = 2 and t' = 1, 12 % between 4 and 3)
    st = tr.optimizer.state
    big_m, big_v = float(max(g['exp_avg_norms'])), float(max(g['exp_avg_sq_norms']))
    errs = []
    for i, n in enumerate(names):
        s_ = st[params[n]]
        assert float(s_['step']) == float(g['state_steps'][i]) == 4.0, (n, float(s_['step']))
        mn, vn = float(g['exp_avg_norms'][i]), float(g['exp_avg_sq_norms'][i])
        if mn > 1e-3 * big_m:
            errs.append(('exp_avg proj', cases.projection_error(n, s_['exp_avg'], g['exp_avg_proj'][i], mn), M_PROJ_TOL, n))
            errs.append(('exp_avg norm', abs(float(s_['exp_avg'].double().norm()) - mn) / mn, M_NORM_TOL, n))
        if vn > 1e-6 * big_v:
            errs.append(('exp_avg_sq proj', cases.projection_error(n, s_['exp_avg_sq'], g['exp_avg_sq_proj'][i], vn), V_PROJ_TOL, n))
            errs.append(('exp_avg_sq norm', abs(float(s_['exp_avg_sq'].double().norm()) - vn) / vn, V_NORM_TOL, n))
    for i, (got, want) in enumerate(zip(step_dn, g['step_delta_norms'])):
        print('step %d: |update| %.6g vs reference %.6g (%.3g)' % (i + 1, got, want, abs(got - want) / want))
    for kind in ('exp_avg proj', 'exp_avg norm', 'exp_avg_sq proj', 'exp_avg_sq norm'):
        top = sorted((e for e in errs if e[0] == kind), key=lambda e: -e[1])[:4]
        print('optimiser state after step 4, largest %s errors: %s' % (kind, ['%.3g %s' % (e[1], e[3]) for e in top]))
    bad = [e for e in errs if not e[1] < e[2]]
    assert not bad, bad
    for i, (got, want) in enumerate(zip(step_dn, g['step_delta_norms'])):
        assert abs(got - want) <= STEP_DN_TOL * want, (i, got, want)


def test_train_steps_reduce_loss():
    """Three optimisation steps (train mode: dropout + BN batch stats; Adam, clip 0.1) on one fixed batch."""
    opt = gvd_amd.opts.default_opt(vocab_size=1000, t_attn_size=10)
    torch.manual_seed(0)
    model = att_model.TopDownModel(opt)
    model.load_state_dict(synth.init_state_dict(opt, seed=0))
    model = model.cuda().train()
    tr = train.Trainer(model, opt)
    inp = synth.trim_to_batch(synth.make_inputs(opt, 8, seed=0, train=True))
    args = synth.as_args(inp, 'cuda')
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    hist = [tr.step(args).cpu() for _ in range(4)]
    assert all(torch.isfinite(h).all() for h in hist)
    assert float(hist[-1][0]) < float(hist[0][0])          # LM loss goes down on the repeated batch
    moved = [n for n, p in model.named_parameters() if not torch.equal(p.detach(), before[n])]
    assert 'core.att_lstm.weight_hh' in moved and 'ctx2pool_grd.0.weight' in moved and 'logit.weight' in moved
    assert 'core.i2h_2.weight' not in moved


def test_sample_after_training_steps_uses_the_updated_weights():
    """main.py's epoch loop validates between training epochs (main.py:690-744: train -> eval on the same model): a
    'sample' call after Trainer.step must decode with the UPDATED parameters.  The inference path keeps re-laid-out
    copies of some weights (att_model._packed, keyed on the parameters' version counters); the own optimiser writes the
    parameters through raw pointers and has to advance those counters itself.  Compared against a freshly constructed
    model that loads the trained state_dict (empty pack cache): ids, log-probabilities and attention logits bit-equal."""
    opt = gvd_amd.opts.default_opt(vocab_size=1000, t_attn_size=10)
    torch.manual_seed(0)
    model = att_model.TopDownModel(opt)
    model.load_state_dict(synth.init_state_dict(opt, seed=3, profile='trained_like'))
    model = model.cuda()
    inp = synth.trim_to_batch(synth.make_inputs(opt, 8, seed=3, train=True))
    args = synth.as_args(inp, 'cuda')
    sargs = [args[i] for i in (0, 4, 3, 7, 9, 10)]          # segs_feat, ppls, num, ppls_feat, sample_idx, pnt_mask

    def decode(m):
        m.eval()
        seq, lps, att2, sim = m._sample(*sargs, {})
        m.check_kernel_status()
        return seq.clone(), lps.clone(), att2.clone(), sim.clone()
    first = decode(model)                                    # fills the pack cache with the initial weights
    opt.learning_rate = 5e-3                                 # (large enough for three steps to change the caption)
    tr = train.Trainer(model, opt)
    model.train()
    for _ in range(3):
        tr.step(args)
    got = decode(model)
    fresh = att_model.TopDownModel(opt)
    fresh.load_state_dict({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    want = decode(fresh.cuda())
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    assert not torch.equal(got[1], first[1])                 # the steps did move the decode


@pytest.mark.parametrize('mode', ['sample', 'train'])
def test_bench_under_torchrun_single_rank(mode):
    """bench.py launched exactly like the driver launches it (torch.distributed.run, one rank per GPU): RCCL
    init, barrier, MAX all-reduce and (train) the bucketed gradient all-reduce all run on this 1-GPU box."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GVD_DP_FORCE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
           '127.0.0.1', '--master-port', '29617', os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '2',
           '--warmup', '1', '--batch', '8', '--vocab', '1000', '--no-cpu-baseline', '--mode', mode]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    j = json.loads(line)
    assert j['n_gpus'] == 1 and j['value'] > 0 and j['steps'] == 2


@pytest.mark.parametrize('mode', ['sample', 'train'])
def test_bench_two_ranks_share_the_one_gpu(mode):
    """The N = 2 path of bench.py end to end on a 1-GPU box: torch.distributed.run with two ranks that share cuda:0 over
    gloo (--dist-backend gloo; RCCL refuses two ranks on one device) - per-rank shards, barrier, the bucketed gradient
    all-reduce from the autograd hooks (train), MAX + per-rank times.  A code-path run, not a measurement."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', '29631', os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2',
           '--warmup', '1', '--batch', '8', '--vocab', '1000', '--no-cpu-baseline', '--mode', mode, '--dist-backend', 'gloo']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert j['n_gpus'] == 2 and j['value'] > 0 and len(j['per_rank_seconds']) == 2
    assert abs(j['ms_per_step'] * 2 / 1e3 - max(j['per_rank_seconds'])) < 1e-3
    # an N > 1 line is as complete as the N = 1 line of the same command: same keys, a measured roofline block (the extra
    # measurement step runs on every rank, collective-safe), cpu_baseline present as a pointer to the N = 1 value
    cmd1 = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--batch', '8',
            '--vocab', '1000', '--mode', mode, '--cpu-seconds', '1']
    out1 = subprocess.run(cmd1, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out1.returncode == 0, out1.stderr[-2000:]
    j1 = json.loads([l for l in out1.stdout.splitlines() if l.startswith('{')][-1])
    cmd2 = [c for c in cmd if c != '--no-cpu-baseline']
    out2 = subprocess.run(cmd2, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out2.returncode == 0, out2.stderr[-2000:]
    j2 = json.loads([l for l in out2.stdout.splitlines() if l.startswith('{')][-1])
    assert set(j2) == set(j1), (sorted(set(j1) ^ set(j2)))
    assert set(j2['config']) == set(j1['config']), (sorted(set(j1['config']) ^ set(j2['config'])))
    assert j2['roofline'] is not None and j2['roofline']['frac'] is not None and j2['roofline']['frac'] > 0
    if mode == 'sample':
        assert j2['roofline_mfma'] is not None and j2['roofline_mfma']['frac'] > 0
    assert j2['cpu_baseline'] is not None and j2['cpu_baseline']['value'] is None and 'n_gpus = 1' in j2['cpu_baseline']['sample']
    assert j1['cpu_baseline']['value'] > 0


@pytest.mark.parametrize('B,R', [(2, 40), (3, 100)])
def test_region_feature_rows_train_forward_and_backward(B, R):
    """ops.region_feature_rows_train (fused row kernel forward + fused backward) against autograd through the ATen form
    of model.py:336-364 (masked class softmax, three F.layer_norm, concat), incl. the direct gradient of the class
    distribution and a fully masked sample."""
    import torch.nn.functional as F
    g = _g(7 + R)
    D1, nl, cpad = 433, 300, 15
    g_pool = torch.relu(torch.randn(B, R, 2048, generator=g)).cuda().requires_grad_(True)
    loc = torch.relu(torch.randn(B, R, nl, generator=g)).cuda().requires_grad_(True)
    logits = (torch.randn(B, R, D1 + cpad, generator=g) * 2).cuda().requires_grad_(True)
    pm = (torch.rand(B, R + 1, generator=g) < 0.3).to(torch.uint8)
    pm[:, 0] = 0
    pm[0, 1:] = 1
    pm = pm.cuda()
    Gout = torch.randn(B, R, 2784, generator=g).cuda()
    Gsim = torch.randn(B, R, D1, generator=g).cuda()
    # reference
    lm = logits[:, :, :D1].masked_fill(pm[:, 1:].bool().unsqueeze(-1), -1e8)
    p = F.softmax(lm, dim=-1)
    ref = torch.cat([F.layer_norm(g_pool, [2048]), F.layer_norm(loc, [nl]), F.layer_norm(p, [D1])], -1)
    ((ref * Gout[:, :, :2781]).sum() + (p * Gsim).sum()).backward()
    rg = [t.grad.clone() for t in (g_pool, loc, logits)]
    for t in (g_pool, loc, logits):
        t.grad = None
    out, sim = ops.region_feature_rows_train(g_pool, loc, logits, pm, D1, pad_to=32)
    assert out.shape == (B, R, 2784) and sim.shape == (B, R, D1)
    np.testing.assert_allclose(out[:, :, :2781].detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-4, atol=2e-5)
    assert float(out[:, :, 2781:].abs().max()) == 0.0
    np.testing.assert_allclose(sim.detach().cpu().numpy(), p.detach().cpu().numpy(), rtol=1e-5, atol=1e-7)
    ((out * Gout).sum() + (sim * Gsim).sum()).backward()
    for t, r, tol in zip((g_pool, loc, logits), rg, (2e-4, 2e-4, 2e-4)):
        a, b = t.grad.cpu().numpy(), r.cpu().numpy()
        assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), (np.abs(a - b).max(), np.abs(b).max())
    assert float(logits.grad[:, :, D1:].abs().max()) == 0.0 and float(logits.grad[0].abs().max()) == 0.0


def test_cls_loss_class_last_layout():
    """gvd_cls_loss through element strides: the transposed view of a class-last tensor gives the same loss and the same
    gradient as the contiguous [B,D1,R] tensor."""
    g = _g(11)
    B, D1, R, K = 3, 433, 200, 5
    p = torch.softmax(torch.randn(B, D1, R, generator=g), 1).cuda()
    tgt = torch.randint(0, D1, (B, K, R), generator=g)
    tgt[torch.rand(B, K, R, generator=g) < 0.7] = 0
    tgt = tgt.cuda()
    a = p.clone().requires_grad_(True)
    la = ops.cls_loss(a, tgt)
    la.backward()
    bt = p.transpose(1, 2).contiguous().requires_grad_(True)      # class-last in memory
    lb = ops.cls_loss(bt.transpose(1, 2), tgt)
    lb.backward()
    assert float(la) == float(lb)
    assert torch.equal(bt.grad.transpose(1, 2), a.grad)


@pytest.mark.parametrize('B,T', [(5, 10), (64, 10), (3, 1), (40, 7)])
def test_gru_train_matches_library_gru(B, T):
    """gru_fn (persistent-kernel forward + hand-scheduled BPTT with gvd_gru_bwd_step) against autograd through the
    native torch GRU on the GPU (eval mode: no inter-layer dropout)."""
    from gvd_amd import gru_fn
    torch.manual_seed(3)
    gru = torch.nn.GRU(1024, 512, 2, dropout=0.2, bidirectional=True, batch_first=True).cuda().eval()
    x = torch.randn(B, T, 1024, device='cuda')
    G = torch.randn(B, T, 1024, device='cuda')
    xr = x.clone().requires_grad_(True)
    with torch.backends.cudnn.flags(enabled=False):
        yr = gru(xr)[0]
        (yr * G).sum().backward()
    ref = {n: p.grad.clone() for n, p in gru.named_parameters()}
    gru.zero_grad(set_to_none=True)
    xm = x.clone().requires_grad_(True)
    flags = []
    ym = gru_fn.gru_bidir_2layer_train(xm, gru, flags=flags)
    np.testing.assert_allclose(ym.detach().cpu().numpy(), yr.detach().cpu().numpy(), rtol=1e-4, atol=2e-5)
    (ym * G).sum().backward()
    assert all(int(f.sum()) == 0 for f in flags)
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-12))
    assert rel(xm.grad, xr.grad) < 1e-4
    for n, p in gru.named_parameters():
        assert rel(p.grad, ref[n]) < 1e-4, (n, rel(p.grad, ref[n]))


def test_batch_dp_8x32_matches_reference_shard_by_shard(golden_dir):
    """BASELINE configs[3] (B=256 as 8 replicas x 32 segments): the reference was run shard by shard and its gradients
    averaged over the shards (oracle/make_golden.py mode 'dp' = nn.DataParallel's loss .sum() / numel() backward).  The HIP
    model runs the same 8 shards — each trimmed on its own, as a rank's loader would hand it over — and accumulates
    grad / 8 = what the all-reduce(sum)/N of dist.GradAllReducer leaves on every rank (tests/test_gpu_dist.py checks that
    the collective path equals this mean): per-shard losses within 1e-4, per-parameter norms of the averaged gradient
    within 2e-3."""
    name = 'dp8x32_v5000_ft10_trained'
    path = os.path.join(golden_dir, name + '.npz')
    if not os.path.exists(path):
        pytest.skip('fixture not generated')
    g = np.load(path)
    opt, sd, inp = cases.build_case(name)
    assert cases.weight_fingerprint(sd) == int(g['weight_fp']) and cases.input_fingerprint(inp) == int(g['input_fp'])
    model = att_model.TopDownModel(opt)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    n = cases.CASES[name]['shards']
    per = cases.CASES[name]['B'] // n
    w = cases.GRAD_WEIGHTS
    for r in range(n):
        sub = synth.trim_to_batch({k: v[r * per:(r + 1) * per].contiguous() for k, v in inp.items()})
        lm, a2, gl, cl = model(*synth.as_args(sub, 'cuda'), 'MLE')
        np.testing.assert_allclose(np.array([float(lm), float(a2), float(gl), float(cl)]), g['shard_losses'][r], atol=1e-4)
        ((lm.sum() + w['w_att2'] * a2.sum() + w['w_grd'] * gl.sum() + w['w_cls'] * cl.sum()) / n).backward()
    ref = dict(zip([str(x) for x in g['grad_names']], g['grad_norms']))
    params = dict(model.named_parameters())
    for pn, want in ref.items():
        got = float(params[pn].grad.double().norm())
        assert abs(got - want) / max(want, 1e-3) < 2e-3, '%s: |grad| %.6g vs reference %.6g' % (pn, got, want)
    names = [str(x) for x in g['grad_names']]
    _check_projections({pn: params[pn].grad for pn in names}, names, g['grad_norms'], g['grad_proj'], what='dp-mean grad')
