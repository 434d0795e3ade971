"""Where the 3 ms of a batch_size = 4 `forward(..., 'sample')` call go (BASELINE configs[1]): device-busy time (sum of kernel
durations) against the wall time per call, the gaps between consecutive kernels, the kernels by total time.
    python tools/b4_timeline.py [B]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402
import gvd_amd  # noqa: E402,F401
from gvd_amd import att_model, opts, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
opt = opts.default_opt(vocab_size=5000, t_attn_size=10)
sd = synth.init_state_dict(opt, seed=5, profile='trained_like')
model = att_model.TopDownModel(opt)
model.load_state_dict(sd)
model = model.cuda().eval()
inp = synth.make_inputs(opt, B, seed=0, train=False)
keys = ('segs_feat', 'ppls', 'num', 'ppls_feat', 'sample_idx', 'pnt_mask')
s4 = {k: inp[k].cuda() for k in keys}
dummy = torch.zeros(B, dtype=torch.uint8, device='cuda')
fwd = (s4['segs_feat'], dummy, dummy, s4['num'], s4['ppls'], dummy, dummy, s4['ppls_feat'], dummy, s4['sample_idx'],
       s4['pnt_mask'], 'sample', {'sample_max': 1, 'beam_size': 1})
with torch.no_grad():
    for _ in range(5):
        model(*fwd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        model(*fwd)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 20
    N = 5
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(N):
            model(*fwd)
        torch.cuda.synchronize()
ks = sorted([e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA], key=lambda e: e.time_range.start)
busy = sum(e.time_range.end - e.time_range.start for e in ks)
gaps = [b.time_range.start - a.time_range.end for a, b in zip(ks[:-1], ks[1:])]
gaps = [g for g in gaps if 0 < g < 200]           # (the gap between calls includes the status read: left out above 200 us)
print('B = %d: wall %.3f ms per call (unprofiled); %d device activities per call, busy %.3f ms per call, %d gaps summing to %.3f ms per call'
      % (B, wall * 1e3, len(ks) // N, busy / N / 1e3, len(gaps) // N, sum(gaps) / N / 1e3))
tot = {}
for e in ks:
    r = tot.setdefault(e.name[:90], [0, 0.0])
    r[0] += 1
    r[1] += e.time_range.end - e.time_range.start
for name, r in sorted(tot.items(), key=lambda kv: -kv[1][1])[:28]:
    print('%8.1f us %4d x  %s' % (r[1] / N, r[0] // N, name))
