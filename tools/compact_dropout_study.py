"""What does the compacted training layout (GVD_TRAIN_COMPACT=1, train_compact.py) change under LIVE dropout?

In eval-mode arithmetic the compacted step equals the full-row step (losses and gradients: the reference goldens, run with
the knob on).  With dropout live the reference draws an independent mask for each of the n masked rows of a segment, the
compacted layout ONE draw for their weighted representative: a different stochastic regulariser.  This tool measures the
difference in the only quantity that reaches the weights - the EXPECTED gradient: N train-mode steps (fresh dropout draws,
same batch, no optimiser update) in each layout, per parameter
    bias   = |mean_full - mean_compact| / |mean_full|
    noise  = sqrt(var_full / N + var_compact / N) summed in quadrature over the elements / |mean_full|   (Monte-Carlo error of
             that difference)
and the same for the four losses.  bias ~ noise means no difference is detectable at this N.
    python tools/compact_dropout_study.py [N draws] [batch] > profiles/r04/compact_dropout_study.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gvd_amd  # noqa: E402,F401
from gvd_amd import att_model, opts, synth, train  # noqa: E402
from oracle import cases  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
opt = opts.default_opt(vocab_size=1000, t_attn_size=10)
for k, v in cases.GRAD_WEIGHTS.items():
    setattr(opt, k, v)
sd = synth.init_state_dict(opt, seed=41, profile='trained_like')
args = synth.as_args(synth.trim_to_batch(synth.make_inputs(opt, B, seed=41, train=True)), 'cuda')
stats = {}
for mode in ('0', '1'):
    os.environ['GVD_TRAIN_COMPACT'] = mode
    model = att_model.TopDownModel(opt)
    model.load_state_dict(sd)
    model = model.cuda().train()
    torch.manual_seed(1234 + int(mode))
    s1, s2, l1, l2 = {}, {}, torch.zeros(4, dtype=torch.float64), torch.zeros(4, dtype=torch.float64)
    for i in range(N):
        model.zero_grad(set_to_none=True)
        losses = model(*args, 'MLE')
        train.combine_losses(losses, opt).backward()
        lv = torch.cat([l.detach() for l in losses]).double().cpu()
        l1 += lv
        l2 += lv * lv
        for n, p in model.named_parameters():
            if p.grad is None:
                continue
            g = p.grad.double()
            if n not in s1:
                s1[n], s2[n] = torch.zeros_like(g), torch.zeros_like(g)
            s1[n] += g
            s2[n] += g * g
    model.check_kernel_status()
    stats[mode] = (s1, s2, l1, l2)
    del model
out = {'draws_per_layout': N, 'batch': B, 'params': {}, 'note': __doc__.split('\n\n')[1]}
f, c = stats['0'], stats['1']
lm_f, lm_c = f[2] / N, c[2] / N
lv_f, lv_c = f[3] / N - lm_f ** 2, c[3] / N - lm_c ** 2
out['losses'] = {'names': ['lm', 'att2', 'grd', 'cls'], 'mean_full': lm_f.tolist(), 'mean_compact': lm_c.tolist(),
                 'mc_error_of_the_difference': torch.sqrt((lv_f + lv_c).clamp_min(0) / N).tolist()}
worst = []
for n in f[0]:
    mf, mc = f[0][n] / N, c[0][n] / N
    vf, vc = (f[1][n] / N - mf * mf).clamp_min(0), (c[1][n] / N - mc * mc).clamp_min(0)
    ref = float(mf.norm())
    if ref == 0:
        continue
    bias = float((mf - mc).norm()) / ref
    noise = float(torch.sqrt(((vf + vc) / N).sum())) / ref
    out['params'][n] = {'bias': bias, 'mc_noise': noise, 'ratio': bias / max(noise, 1e-30)}
    worst.append((bias / max(noise, 1e-30), n))
worst.sort(reverse=True)
out['largest_bias_over_noise'] = [{'param': n, 'ratio': r, **out['params'][n]} for r, n in worst[:8]]
rs = torch.tensor([v['ratio'] for v in out['params'].values()])
out['summary'] = {'params': len(rs), 'median_ratio': float(rs.median()), 'max_ratio': float(rs.max()),
                  'params_with_ratio_above_2': int((rs > 2).sum())}
print(json.dumps(out, indent=1))
