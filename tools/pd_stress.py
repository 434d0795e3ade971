"""Stress the persistent kernels (decode B<=4, GRU) for rare barrier / visibility problems: many back-to-back calls,
varying batch sizes, serial and two-stream pipelined; every call must reproduce its first result bit for bit and no
barrier may time out.  python tools/pd_stress.py [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gvd_amd
from gvd_amd import att_model, ops, synth
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
keys = ('segs_feat', 'ppls', 'num', 'ppls_feat', 'sample_idx', 'pnt_mask')
bad = 0
for Ft in (10, 480):
    opt = gvd_amd.opts.default_opt(vocab_size=5000, t_attn_size=Ft)
    sd = synth.init_state_dict(opt, seed=Ft, profile='trained_like')
    model = att_model.TopDownModel(opt); model.load_state_dict(sd); model = model.cuda().eval()
    batches = []
    for B in (1, 2, 3, 4):
        inp = synth.make_inputs(opt, B, seed=B, train=False)
        batches.append([inp[k].cuda() for k in keys])
    with torch.no_grad():
        ref = [model._sample(*b) for b in batches]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        statuses = []
        for i in range(iters):
            b = i % 4
            out = model._sample(*batches[b])
            statuses.append(ops.greedy_decode.last_status)
            if not (torch.equal(out[0], ref[b][0]) and torch.equal(out[2], ref[b][2])):
                bad += 1
        for i in range(iters // 8):
            outs = model.sample_pipelined(batches)
            for o, r in zip(outs, ref):
                if not (torch.equal(o[0], r[0]) and torch.equal(o[2], r[2])):
                    bad += 1
        torch.cuda.synchronize()
        to = sum(int(s) for s in statuses)
        print('Ft=%d: %d serial + %d pipelined calls in %.1f s, mismatches %d, barrier timeouts %d'
              % (Ft, iters, iters // 8 * 4, time.perf_counter() - t0, bad, to))
        bad += to
print('STRESS', 'OK' if bad == 0 else 'FAILED')
sys.exit(1 if bad else 0)
