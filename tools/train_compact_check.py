"""First thing to run on a device for the compacted training layout (grounded-video-description_amd/train_compact.py,
GVD_TRAIN_COMPACT=1; its maths is pinned on the CPU by tests/test_train_compact_cpu.py, its HIP side - the key-bias
operand of the encoder's softmax row kernel, the plumbing - has not run on a GPU yet):
  1. the reference case mle_b64_v5000_ft10_trained on the full row set and on the compacted layout: the four losses (vs the
     reference's) and every parameter gradient, compacted vs full (relative Frobenius error, worst parameter);
  2. the batch_size = 64 training step timed both ways in this process.
    python tools/train_compact_check.py [case] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import gvd_amd  # noqa: E402,F401
from gvd_amd import att_model, opts, synth, train  # noqa: E402
from oracle import cases  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'mle_b64_v5000_ft10_trained'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', name + '.npz'))
opt, sd, inp = cases.build_case(name)
args = synth.as_args(inp, 'cuda')
w = cases.GRAD_WEIGHTS
res = {}
for mode in ('0', '1'):
    os.environ['GVD_TRAIN_COMPACT'] = mode
    model = att_model.TopDownModel(opt)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    lm, a2, gl, cl = model(*args, 'MLE')
    (lm.sum() + w['w_att2'] * a2.sum() + w['w_grd'] * gl.sum() + w['w_cls'] * cl.sum()).backward()
    model.check_kernel_status()
    res[mode] = (np.array([float(lm), float(a2), float(gl), float(cl)]),
                 {n: p.grad.detach().double().cpu() for n, p in model.named_parameters() if p.grad is not None})
    print('GVD_TRAIN_COMPACT=%s losses %s  |delta vs reference| %.2e' % (mode, res[mode][0], np.abs(res[mode][0] - g['losses']).max()),
          flush=True)
worst = (0.0, None)
gmax = max(float(v.norm()) for v in res['0'][1].values())
for n, a in res['0'][1].items():
    b = res['1'][1][n]
    if float(a.norm()) > 1e-6 * gmax:
        worst = max(worst, (float((a - b).norm() / a.norm()), n))
print('worst relative gradient difference compacted vs full: %.3g (%s)' % worst, flush=True)

if steps <= 0:
    sys.exit(0)
opt = opts.default_opt(vocab_size=5000, t_attn_size=10)
model = att_model.TopDownModel(opt)
model.load_state_dict(synth.init_state_dict(opt, seed=0, profile='trained_like'))
model = model.cuda().train()
a = synth.as_args(synth.trim_to_batch(synth.make_inputs(opt, 64, seed=200, train=True)), 'cuda')
tr = train.Trainer(model, opt)
for r in range(2):
    for mode in ('0', '1'):
        os.environ['GVD_TRAIN_COMPACT'] = mode
        tr.step(a); tr.step(a)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step(a)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print('round %d GVD_TRAIN_COMPACT=%s: %.3f ms per step (%.1f segments/s)' % (r, mode, 1e3 * dt, 64 / dt), flush=True)
