"""Interleaved A/B of the training step in ONE process (same clocks, same allocator state): the fused elementwise kernels +
own clip/Adam with the status read after the optimiser (default), the same with the read first, and the ATen passes +
torch's fused Adam.
    python tools/train_ab.py [rounds] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gvd_amd  # noqa: E402,F401
from gvd_amd import att_model, ops, opts, synth, train  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
opt = opts.default_opt(vocab_size=5000, t_attn_size=10)
model = att_model.TopDownModel(opt)
model.load_state_dict(synth.init_state_dict(opt, seed=0, profile='trained_like'))
model = model.cuda().train()
a = synth.as_args(synth.trim_to_batch(synth.make_inputs(opt, 64, seed=200, train=True)), 'cuda')
os.environ['GVD_OWN_ADAM'] = '1'
tr_own = train.Trainer(model, opt)
os.environ['GVD_OWN_ADAM'] = '0'
tr_torch = train.Trainer(model, opt)
VARIANTS = [('fused+own-adam, read after optimiser', tr_own, True, '0'), ('fused+own-adam, read first', tr_own, True, '1'),
            ('aten+torch-adam', tr_torch, False, '0')]
for name, tr, fused, sync_first in VARIANTS:        # warm-up of every variant (allocator, bucket discovery)
    ops.FUSED_TRAIN_ELEMENTWISE = fused
    os.environ['GVD_TRAIN_DEFER_STATUS'] = '0' if sync_first == '1' else '1'
    tr.step(a); tr.step(a)
res = {n: [] for n, _, _, _ in VARIANTS}
for r in range(rounds):
    for name, tr, fused, sync_first in VARIANTS:
        ops.FUSED_TRAIN_ELEMENTWISE = fused
        os.environ['GVD_TRAIN_DEFER_STATUS'] = '0' if sync_first == '1' else '1'
        tr.step(a)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step(a)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / steps * 1e3)
for name, v in res.items():
    print('%-40s ms/step per round: %s   min %.3f  (%.1f segments/s)' % (name, ' '.join('%.3f' % x for x in v), min(v), 64e3 / min(v)),
          flush=True)
