"""How far is the REFERENCE's fp32 gradient (tests/golden/mle_*.npz: norms + seeded projections) from the fp64 value of the same
graph?  CPU only: the oracle's 'MLE' forward + autograd backward in fp32 and in fp64 on a reference case, per parameter the
projection error against the reference fixture and |g32 - g64| / |g64|.  One ReLU pre-activation falling on the other side of
zero in fp32 moves a Linear layer's weight gradient by ~1e-2 relative - the bound of tests/test_gpu_train.py's direction check.
    python tools/reference_grad_noise.py <case> [substring of the parameter names to print]"""
import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gvd_amd
from gvd_amd import synth
from oracle import cases, gvd_oracle as O
torch.set_num_threads(8)
name = sys.argv[1]
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', '%s.npz' % name))
opt, sd, inp = cases.build_case(name)
w = cases.GRAD_WEIGHTS
def grads(dtype):
    W = {k: (v.to(dtype).clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v) for k, v in sd.items()}
    if dtype == torch.float64:
        for k in ('att_embed_aux.0.running_mean','att_embed_aux.0.running_var'): W[k] = W[k].double()
    a = [inp[k].to(dtype) if inp[k].is_floating_point() else inp[k] for k in synth.FORWARD_ORDER]
    lm, a2, gl, cl, _ = O.forward_train(W, opt, *a)
    (lm + w['w_att2']*a2 + w['w_grd']*gl + w['w_cls']*cl).backward()
    return {k: v.grad for k, v in W.items() if getattr(v, 'grad', None) is not None}
g32 = grads(torch.float32)
torch.set_default_dtype(torch.float64)
g64 = grads(torch.float64)
torch.set_default_dtype(torch.float32)
names = [str(n) for n in g['grad_names']]
for n, wn, wp in zip(names, g['grad_norms'], g['grad_proj']):
    if (len(sys.argv) > 2 and sys.argv[2] in n) or (len(sys.argv) <= 2 and ('ctx2pool_grd' in n or 'vis_embed' in n)):
        e32 = cases.projection_error(n, g32[n].float(), wp, wn)
        e64 = cases.projection_error(n, g64[n].float(), wp, wn)
        rel = float((g32[n].double()-g64[n]).norm()/g64[n].norm())
        print(n, 'ref norm %.4g' % wn, 'oracle32 vs ref %.3g' % e32, 'oracle64 vs ref %.3g' % e64, '|g32-g64|/|g64| %.3g' % rel)
