"""Which fused training component costs gradient accuracy?  Runs one 'MLE' edge case (oracle/edge_cases.py) on the HIP path
under several knob settings and prints, per setting, the parameters with the largest elementwise relative gradient error
against the CPU oracle's autograd (itself within 1e-5 of an fp64 run).   python tools/grad_diag.py [case]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gvd_amd  # noqa: E402,F401
from gvd_amd import att_model, synth  # noqa: E402
from oracle import cases, edge_cases, gvd_oracle as O  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'train_masked_frame'
opt, sd, inp = edge_cases.TRAIN_EDGE_CASES[name]()
w = cases.GRAD_WEIGHTS
W = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v) for k, v in sd.items()}
olm, oa2, ogl, ocl, _ = O.forward_train(W, opt, *[inp[k] for k in synth.FORWARD_ORDER])
(olm + w['w_att2'] * oa2 + w['w_grd'] * ogl + w['w_cls'] * ocl).backward()
gmax = max(float(v.grad.norm()) for v in W.values() if torch.is_tensor(v) and v.grad is not None)

SETTINGS = [
    ('default', {}),
    ('head pad 192', {'GVD_TRAIN_HEAD_PAD': '192'}),
    ('library attention core', {'GVD_ENC_TRAIN_MFMA': '0'}),
    ('autograd LayerNorm', {'GVD_LN_FUSED_BWD': '0'}),
    ('library attn + autograd LN', {'GVD_ENC_TRAIN_MFMA': '0', 'GVD_LN_FUSED_BWD': '0'}),
    ('unfused P5', {'GVD_P5_FUSED_TRAIN': '0'}),
    ('library GRU', {'GVD_GRU_TRAIN': '0'}),
]
for label, env in SETTINGS:
    for k in ('GVD_TRAIN_HEAD_PAD', 'GVD_ENC_TRAIN_MFMA', 'GVD_LN_FUSED_BWD', 'GVD_P5_FUSED_TRAIN', 'GVD_GRU_TRAIN'):
        os.environ.pop(k, None)
    os.environ.update(env)
    model = att_model.TopDownModel(opt)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    lm, a2, gl, cl = model(*synth.as_args(inp, 'cuda'), 'MLE')
    (lm.sum() + w['w_att2'] * a2.sum() + w['w_grd'] * gl.sum() + w['w_cls'] * cl.sum()).backward()
    rows = []
    for n, p in model.named_parameters():
        if W[n].grad is None or p.grad is None:
            continue
        want = W[n].grad.double()
        wn = float(want.norm())
        if wn <= 1e-6 * gmax:
            continue
        rows.append((float((p.grad.double().cpu() - want).norm()) / wn, n))
    rows.sort(reverse=True)
    print('%-28s losses d=%.2e | ' % (label, max(abs(float(a) - float(b)) for a, b in zip((lm, a2, gl, cl), (olm, oa2, ogl, ocl))))
          + '  '.join('%.2e %s' % (e, n.replace('obj_interact.encoder.layers.', 'enc.')) for e, n in rows[:5]), flush=True)
