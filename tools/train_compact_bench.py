"""The batch_size = 64 training step on the compacted training layout (GVD_TRAIN_COMPACT=1, train_compact.py) next to the
full row set, in ONE process, as one JSON line - run by bench.py's train section in a SUBPROCESS (a side measurement of an
opt-in path must not be able to take the benchmark line down) and usable on its own:
    python tools/train_compact_bench.py [steps]
Weights + inputs = the committed reference case mle_b64_v5000_ft10_trained (regenerated from its seeds, as bench.py's
section does); `parity` = its four eval-mode losses on the compacted layout against the reference's own (tests/golden) and
the worst relative parameter-gradient difference compacted vs full rows."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import gvd_amd  # noqa: E402,F401
from gvd_amd import att_model, opts, synth, train, train_compact  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W = dict(w_att2=0.05, w_grd=0.3, w_cls=0.1)          # the loss weights of the gradient fixtures
opt = opts.default_opt(vocab_size=5000, t_attn_size=10)
sd = synth.init_state_dict(opt, seed=5, profile='trained_like')
a = synth.as_args(synth.trim_to_batch(synth.make_inputs(opt, 64, seed=5, train=True)), 'cuda')
gpath = os.path.join(ROOT, 'tests', 'golden', 'mle_b64_v5000_ft10_trained.npz')
out = {'batch': 64, 'rows_full': int(a[4].shape[1])}
c = train_compact.compact_regions(a[4], a[7], a[10], a[8])            # ppls, ppls_feat, pnt_mask, frm_mask
out['rows_per_segment'] = None if c is None else int(c['Rc'])
res = {}
for mode in ('0', '1'):
    os.environ['GVD_TRAIN_COMPACT'] = mode
    model = att_model.TopDownModel(opt)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    lm, a2, gl, cl = model(*a, 'MLE')
    (lm.sum() + W['w_att2'] * a2.sum() + W['w_grd'] * gl.sum() + W['w_cls'] * cl.sum()).backward()
    model.check_kernel_status()
    res[mode] = (np.array([float(t.detach()) for t in (lm, a2, gl, cl)]),
                 {n: p.grad.detach().double() for n, p in model.named_parameters() if p.grad is not None})
    del lm, a2, gl, cl
gmax = max(float(v.norm()) for v in res['0'][1].values())
worst = max((float((g - res['1'][1][n]).norm() / g.norm()), n) for n, g in res['0'][1].items() if float(g.norm()) > 1e-6 * gmax)
out['parity'] = {'max_abs_loss_diff_vs_full_rows': float(np.abs(res['1'][0] - res['0'][0]).max()),
                 'worst_rel_grad_diff_vs_full_rows': worst[0], 'worst_grad': worst[1]}
if os.path.exists(gpath):
    d = float(np.abs(res['1'][0] - np.load(gpath)['losses']).max())
    out['parity'].update(max_abs_loss_diff_vs_reference=d, within_1e_4=bool(d <= 1e-4))
del res
model.train()
tr = train.Trainer(model, opt)
for mode, key in (('0', 'full_rows'), ('1', 'compacted_rows'), ('0', 'full_rows_again'), ('1', 'compacted_rows_again')):
    os.environ['GVD_TRAIN_COMPACT'] = mode
    tr.step(a); tr.step(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out[key] = {'ms_per_step': round(1e3 * dt, 3), 'segments_per_s': round(64 / dt, 1)}
out['fell_back_to_full_rows'] = bool(getattr(model, '_train_compact_off', False))
out['steps_timed'] = steps
print('COMPACT_JSON ' + json.dumps(out), flush=True)
