"""Forward-activation accuracy of the training preamble on the HIP path against the CPU oracle (fp32 and fp64 runs of the same
restatement): relative Frobenius error per tensor.   python tools/fwd_diag.py [case]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gvd_amd  # noqa: E402,F401
from gvd_amd import att_model, synth  # noqa: E402
from oracle import edge_cases, gvd_oracle as O  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'train_masked_frame'
opt, sd, inp = edge_cases.TRAIN_EDGE_CASES[name]()
keys = ('segs_feat', 'num', 'ppls', 'ppls_feat', 'sample_idx', 'pnt_mask')


def oracle(dt):
    torch.set_default_dtype(dt)
    W = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd.items()}
    a = [inp[k].to(dt) if inp[k].is_floating_point() else inp[k] for k in keys]
    with torch.no_grad():
        pre = O.preamble(W, opt, *a)
    torch.set_default_dtype(torch.float32)
    return pre


p32, p64 = oracle(torch.float32), oracle(torch.float64)
model = att_model.TopDownModel(opt)
model.load_state_dict(sd)
model = model.cuda().eval()
d = {k: inp[k].cuda() for k in keys}
res = {}
for label, grad in (('train path (grad on)', True), ('inference dense', False)):
    with torch.set_grad_enabled(grad):
        pre = model._preamble(d['segs_feat'], d['num'], d['ppls'], d['ppls_feat'], d['sample_idx'], d['pnt_mask'])
    res[label] = pre
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
print('%-16s %12s %12s %12s' % ('tensor', 'cpu32 vs 64', 'hip(train)', 'hip(infer)'))
for k in ('fc', 'g_pool', 'sim_mat_static', 'pool', 'p_pool', 'conv', 'p_conv'):
    print('%-16s %12.3e %12.3e %12.3e' % (k, rel(p32[k], p64[k]), rel(res['train path (grad on)'][k].detach().cpu(), p64[k]),
                                          rel(res['inference dense'][k].detach().cpu(), p64[k])))
