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


GRAD_CASES = [n for n, s in cases.CASES.items() if s['mode'] == 'MLE' and s['B'] <= 8]


@pytest.mark.parametrize('name', GRAD_CASES)
def test_mle_gradients_match_reference(name, golden_dir):
    """Eval-mode (dropout off, BN running stats) gradients of lm + w_att2*att2 + w_grd*grd + w_cls*cls: every
    parameter's gradient L2 norm vs the reference's (oracle/make_golden.py), plus the 4 losses."""
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    opt, sd, inp = cases.build_case(name)
    model = att_model.TopDownModel(opt)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    lm, a2, gl, cl = model(*synth.as_args(inp, 'cuda'), 'MLE')
    np.testing.assert_allclose(np.array([float(lm), float(a2), float(gl), float(cl)]), g['losses'], atol=1e-4)
    w = cases.GRAD_WEIGHTS
    (lm.sum() + w['w_att2'] * a2.sum() + w['w_grd'] * gl.sum() + w['w_cls'] * cl.sum()).backward()
    ref = dict(zip([str(n) for n in g['grad_names']], g['grad_norms']))
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
    for n in ('core.i2h_2.weight', 'core.h2h_2.weight'):
        assert params[n].grad is None or float(params[n].grad.abs().sum()) == 0.0
    print('worst relative grad-norm error', worst)


@pytest.mark.parametrize('name', sorted(edge_cases.TRAIN_EDGE_CASES))
def test_mle_edge_shapes_match_oracle(name):
    """Training edge shapes (oracle/edge_cases.py: one segment, sizes no tile divides, an annotated frame with every
    proposal masked): 4 losses within 1e-4 of the oracle (itself pinned to the reference on these cases by
    tests/test_oracle_vs_reference.py), every parameter's gradient norm vs the oracle's autograd."""
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
    for n, p in model.named_parameters():
        want = 0.0 if W[n].grad is None else float(W[n].grad.double().norm())
        got = 0.0 if p.grad is None else float(p.grad.double().norm())
        assert abs(got - want) / max(want, 1e-3) < 2e-3, '%s: |grad| %.6g vs oracle %.6g' % (n, got, want)


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
    for n in ('core.i2h_2.weight', 'core.h2h_2.weight'):
        assert torch.equal(params[n].detach(), before[n])


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
