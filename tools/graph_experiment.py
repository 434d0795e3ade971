"""Experiment: capture model._sample (preamble + greedy loop) into a hipGraph via torch.cuda.CUDAGraph and compare
replay latency with eager launches.  python tools/graph_experiment.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gvd_amd
from gvd_amd import att_model, opts, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
opt = opts.default_opt(vocab_size=5000, t_attn_size=10)
sd = synth.init_state_dict(opt, seed=0, profile='trained_like')
model = att_model.TopDownModel(opt); model.load_state_dict(sd); model = model.cuda().eval()
inp = synth.make_inputs(opt, B, seed=0, train=False)
keys = ('segs_feat', 'ppls', 'num', 'ppls_feat', 'sample_idx', 'pnt_mask')
static = [inp[k].cuda() for k in keys]

def run():
    with torch.no_grad():
        return model._sample(*static)

def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

ref = run()
print('eager  ms/call: %.3f' % timeit(run))
# split timing: preamble vs decode
def pre_only():
    with torch.no_grad():
        return model._preamble(static[0], static[2], static[1], static[3], static[4], static[5])
print('preamble eager ms: %.3f' % timeit(pre_only))
try:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): run()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = run()
    torch.cuda.synchronize()
    print('graph  ms/call: %.3f' % timeit(g.replay))
    g.replay(); torch.cuda.synchronize()
    print('graph output == eager:', torch.equal(out[0], ref[0]), torch.equal(out[2], ref[2]))
except Exception as e:
    print('whole-call capture failed:', repr(e)[:300])
