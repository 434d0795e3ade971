"""Debug aid: greedy decode at B<=4 through the persistent kernel vs the multi-kernel loop, step by step.
python tools/pd_debug.py [B] [V] [Ft]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gvd_amd
from gvd_amd import att_model, ops, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
V = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
Ft = int(sys.argv[3]) if len(sys.argv) > 3 else 10
opt = gvd_amd.opts.default_opt(vocab_size=V, t_attn_size=Ft)
sd = synth.init_state_dict(opt, seed=3, profile='trained_like')
model = att_model.TopDownModel(opt); model.load_state_dict(sd); model = model.cuda().eval()
inp = synth.make_inputs(opt, B, seed=5, train=False)
args = [inp[k].cuda() for k in ('segs_feat', 'ppls', 'num', 'ppls_feat', 'sample_idx', 'pnt_mask')]
res = {}
with torch.no_grad():
    for mode in ('0', '1'):
        os.environ['GVD_PERSISTENT'] = mode
        out = model._sample(*args)
        torch.cuda.synchronize()
        print('mode', mode, 'status', int(ops.greedy_decode.last_status))
        res[mode] = [o.cpu() for o in out]
        for _ in range(3): model._sample(*args)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): model._sample(*args)
        torch.cuda.synchronize(); print('  %.3f ms / call' % ((time.perf_counter() - t0) / 20 * 1e3))
a, b = res['0'], res['1']
print('seq equal', torch.equal(a[0], b[0]))
print(a[0][:, :10]); print(b[0][:, :10])
print('lp maxdiff', (a[1] - b[1]).abs().max().item())
print('att2 maxdiff', (a[2] - b[2]).abs().max().item(), 'first step', (a[2][:, 0] - b[2][:, 0]).abs().max().item())
# phase timing of the persistent kernel (workgroup 0 stamps, 100 MHz clock)
os.environ['GVD_PERSISTENT'] = '1'; os.environ['GVD_PD_TRACE'] = '1'
with torch.no_grad():
    model._sample(*args); torch.cuda.synchronize()
tr = ops.greedy_decode.last_trace.cpu().numpy().astype('float64')
L = opt.seq_length
d = (tr[1:] - tr[:-1]).reshape(L, 7) * 0.01      # us
names = ['P1 att-lstm', 'P2 queries', 'P3 attention', 'P4 combine', 'P5 lang-lstm', 'P6 logits', 'P7 token']
print('per-phase us (median over steps, incl. the closing barrier):')
import numpy as np
for i, n in enumerate(names):
    print('  %-14s %.2f' % (n, np.median(d[:, i])))
print('  step total %.2f us;  loop total %.1f us' % (np.median(d.sum(1)), (tr[-1] - tr[0]) * 0.01))
