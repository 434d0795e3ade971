"""One process = one setting of the GVD_* A/B knobs (several of them are read once per process): decode / beam / train a
committed reference case with the environment it was started in and compare with the reference output.
    python tools/knob_check.py decode|beam|train          (exit code 0 = equal)"""
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
else:
    raise SystemExit('mode?')
print('knob_check %s ok: %s' % (mode, {k: v for k, v in os.environ.items() if k.startswith('GVD_')}))
