"""Which torch-native kernels (copies, fills, adds, reductions: everything that is not a kernel of libgvd_hip.so) run inside
the batch_size = 64 training step or the beam = 5 decode, and which Python line issues them?  torch.profiler over ONE call,
grouped by aten op + input shapes + the innermost gvd_amd source line.
    python tools/native_op_profile.py train|beam"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402
import gvd_amd  # noqa: E402,F401
from gvd_amd import att_model, opts, synth, train  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else 'train'
if what == 'train':
    opt = opts.default_opt(vocab_size=5000, t_attn_size=10)
    sd = synth.init_state_dict(opt, seed=5, profile='trained_like')
    a = synth.as_args(synth.trim_to_batch(synth.make_inputs(opt, 64, seed=5, train=True)), 'cuda')
    model = att_model.TopDownModel(opt)
    model.load_state_dict(sd)
    model = model.cuda().train()
    tr = train.Trainer(model, opt)
    run = lambda: tr.step(a)
else:
    opt = opts.default_opt(vocab_size=5000, t_attn_size=10, num_sampled_frm=20)
    sd = synth.init_state_dict(opt, seed=9, profile='trained_like')
    inp = synth.make_inputs(opt, 64, seed=100, train=False)
    d = [inp[k].cuda() for k in ('segs_feat', 'ppls', 'num', 'ppls_feat', 'sample_idx', 'pnt_mask')]
    model = att_model.TopDownModel(opt)
    model.load_state_dict(sd)
    model = model.cuda().eval()

    def run():
        with torch.no_grad():
            model._sample(*d, {'beam_size': 5})
for _ in range(2):
    run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    run()
    torch.cuda.synchronize()
rows = {}
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith('aten::'):
        continue
    dt = sum(k.duration for k in e.kernels)          # device time of the kernels this op launched itself (us)
    if dt <= 0:
        continue
    src = next((s for s in e.stack if 'gvd_amd' in s or 'grounded-video' in s), e.stack[0] if e.stack else '?')
    src = src.split('/')[-1]
    key = (e.name, str(e.input_shapes)[:90], src[:70])
    r = rows.setdefault(key, [0, 0.0])
    r[0] += 1
    r[1] += dt
tot = sum(r[1] for r in rows.values())
print('%s: torch-native device time %.3f ms in %d launches' % (what, tot / 1e3, sum(r[0] for r in rows.values())))
for key, r in sorted(rows.items(), key=lambda kv: -kv[1][1])[:45]:
    print('%8.1f us %4d x  %-28s %-90s %s' % (r[1], r[0], key[0], key[1], key[2]))
