"""Host-side profile of the B=4 `'sample'` call (BASELINE configs[1]): wall per call vs cProfile of the Python launch path."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gvd_amd
from gvd_amd import att_model, opts, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
opt = opts.default_opt(vocab_size=5000, t_attn_size=10)
sd = synth.init_state_dict(opt, seed=0, profile='trained_like')
model = att_model.TopDownModel(opt); model.load_state_dict(sd); model = model.cuda().eval()
inp = synth.make_inputs(opt, B, seed=0, train=False)
a = [inp[k].cuda() for k in ('segs_feat', 'ppls', 'num', 'ppls_feat', 'sample_idx', 'pnt_mask')]
def run(n):
    with torch.no_grad():
        for _ in range(n):
            model._sample(*a)
    torch.cuda.synchronize()
run(5)
t0 = time.perf_counter(); run(50); dt = (time.perf_counter() - t0) / 50
print('wall per call: %.3f ms' % (dt * 1e3))
# host-only: time to ENQUEUE (no sync inside)
torch.cuda.synchronize(); t0 = time.perf_counter()
with torch.no_grad():
    for _ in range(50):
        model._sample(*a)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('enqueue per call: %.3f ms; drain after: %.3f ms' % ((t1 - t0) / 50 * 1e3, (t2 - t1) * 1e3))
pr = cProfile.Profile(); pr.enable(); run(50); pr.disable()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(28)
