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
