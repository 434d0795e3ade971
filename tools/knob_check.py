"""One process = one setting of the GVD_* runtime knobs (several of them are read once per process): decode / beam / train a
committed reference case with the environment it was started in and compare with the reference output.
    python tools/knob_check.py decode|beam|train|timeout_decode|timeout_train          (exit code 0 = equal)
timeout_*: run with GVD_SPIN_LIMIT=1 - the persistent kernels' grid barriers time out at once; the public entry points must
switch those kernels off, recompute, warn once and still deliver the reference result."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import gvd_amd  # noqa: E402,F401
from gvd_amd import att_model, synth  # noqa: E402
from oracle import cases  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
KEYS = ('segs_feat', 'ppls', 'num', 'ppls_feat', 'sample_idx', 'pnt_mask')


def model_for(name):
    g = np.load(os.path.join(GOLD, name + '.npz'))
    opt, sd, inp = cases.build_case(name)
    m = att_model.TopDownModel(opt)
    m.load_state_dict(sd)
    return g, opt, inp, m.cuda().eval()


mode = sys.argv[1]
if mode == 'decode':
    for name in ('greedy_b4_v1000_ft10_trained', 'greedy_b16_v5000_ft10_trained', 'greedy_b4_v5000_ft480_trained'):
        g, opt, inp, m = model_for(name)
        with torch.no_grad():
            seq, lps, att2, sim = m._sample(*[inp[k].cuda() for k in KEYS])
        m.check_kernel_status()
        assert np.array_equal(seq.cpu().numpy(), g['seq']), name
        idx = att_model.attended_region_indices(att2, opt.num_sampled_frm, opt.num_prop_per_frm).cpu().numpy()
        assert np.array_equal(idx, g['att_idx'].astype(np.int64)), name
elif mode == 'beam':
    name = 'beam5_b8_v5000_ft10_t20'
    g, opt, inp, m = model_for(name)
    with torch.no_grad():
        seq, lps, att2, _ = m._sample(*[inp[k].cuda() for k in KEYS], {'beam_size': cases.CASES[name]['K']})
    assert np.array_equal(seq.cpu().numpy(), g['seq']) and np.array_equal(att2.cpu().numpy(), g['att2'].astype(np.int64))
elif mode == 'train':
    name = 'mle_b4_v1000_ft10_trained'
    g, opt, inp, m = model_for(name)
    lm, a2, gl, cl = m(*synth.as_args(inp, 'cuda'), 'MLE')
    np.testing.assert_allclose(np.array([float(lm), float(a2), float(gl), float(cl)]), g['losses'], atol=1e-4)
    w = cases.GRAD_WEIGHTS
    (lm.sum() + w['w_att2'] * a2.sum() + w['w_grd'] * gl.sum() + w['w_cls'] * cl.sum()).backward()
    params = dict(m.named_parameters())
    gmax = float(max(g['grad_norms']))
    for n, want, proj in zip([str(x) for x in g['grad_names']], g['grad_norms'], g['grad_proj']):
        got = float(params[n].grad.double().norm())
        assert abs(got - want) / max(want, 1e-3) < 2e-3, n
        if want > 1e-6 * gmax:
            assert cases.projection_error(n, params[n].grad, proj, want) < 5e-3, n
elif mode == 'timeout_decode':
    import warnings
    from gvd_amd import ops
    assert os.environ.get('GVD_SPIN_LIMIT') == '1'
    # batch_size = 4 (persistent decoder + persistent GRU) and the reference-default 480 frames (long GRU recurrence)
    for name in ('greedy_b4_v1000_ft10_trained', 'greedy_b4_v5000_ft480_trained'):
        g, opt, inp, m = model_for(name)
        ops._persistent.update(on=True, timeouts=0)
        with warnings.catch_warnings(record=True) as w, torch.no_grad():
            warnings.simplefilter('always')
            seq, att2, sim = m(*synth.as_args(inp, 'cuda'), 'sample', {})
        assert any('grid-barrier timeout' in str(x.message) for x in w), 'no timeout was provoked (GVD_SPIN_LIMIT ignored?)'
        assert not ops.persistent_kernels_enabled() and ops._persistent['timeouts'] >= 1
        assert np.array_equal(seq.cpu().numpy(), g['seq']), name
        idx = att_model.attended_region_indices(att2, opt.num_sampled_frm, opt.num_prop_per_frm).cpu().numpy()
        assert np.array_equal(idx, g['att_idx'].astype(np.int64)), name
        # ... and the next call runs without the persistent kernels from the start: no second warning
        with warnings.catch_warnings(record=True) as w, torch.no_grad():
            warnings.simplefilter('always')
            seq2, _, _ = m(*synth.as_args(inp, 'cuda'), 'sample', {})
        assert not any('grid-barrier timeout' in str(x.message) for x in w) and torch.equal(seq2, seq)
elif mode == 'timeout_train':
    import warnings
    from gvd_amd import ops, train
    assert os.environ.get('GVD_SPIN_LIMIT') == '1'
    name = 'step_b4_v1000_ft10_trained'
    g, opt, inp, m = model_for(name)
    for k, v in cases.GRAD_WEIGHTS.items():
        setattr(opt, k, v)
    tr = train.Trainer(m, opt)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        losses = tr.step(synth.as_args(inp, 'cuda')).cpu().numpy()
    assert any('grid-barrier timeout' in str(x.message) for x in w), 'no timeout was provoked (GVD_SPIN_LIMIT ignored?)'
    assert not ops.persistent_kernels_enabled()
    np.testing.assert_allclose(losses, g['losses'], atol=1e-4)
    assert abs(tr.last_grad_norm - float(g['total_grad_norm'])) / float(g['total_grad_norm']) < 2e-3
    assert all(float(st['step']) == 1 for st in tr.optimizer.state.values())      # the invalid attempt updated nothing
else:
    raise SystemExit('mode?')
print('knob_check %s ok: %s' % (mode, {k: v for k, v in os.environ.items() if k.startswith('GVD_')}))
