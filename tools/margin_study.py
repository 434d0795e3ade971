"""Argmax-margin study (SURVEY.md section 7 "hard parts"): how close do the decisions of the greedy decode come to a tie?

The parity contract is bit-exact token ids and per-frame attended-region indices (main.py:364-365) - both are argmax
decisions over fp32 values the HIP path computes in another summation order and with another tanh (csrc/gvd_common.h
tanh_fast: <= 2.5e-7 absolute).  A decision can only flip where the top-1 / top-2 gap is of the order of that noise.
Measured on the CPU oracle (pinned bit-for-bit to the reference):
  * token decisions: gap between the two largest log-probabilities of every step (the UNK rule takes the runner-up when
    the winner is UNK: then the gap between 2nd and 3rd);
  * region decisions: per (segment, step, frame) the gap between the two largest masked attention logits of the frame's
    100 proposals (frames whose proposals are all masked hold 100 equal values: argmax = lowest index on both sides).
Weights: a committed reference case (synthetic `trained_like` profile) AND weights after N real optimisation steps
(oracle 'MLE' forward + autograd + clip 0.1 + Adam 5e-4 on fresh synthetic batches, from torch's default initialisation).
    python tools/margin_study.py [case] [train_steps] [train_batch] [decode_batch] > profiles/r04/margin_study.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gvd_amd  # noqa: E402,F401
from gvd_amd import opts, synth  # noqa: E402
from oracle import cases, gvd_oracle as O  # noqa: E402

torch.set_num_threads(int(os.environ.get('MARGIN_THREADS', '6')))
THRESH = (1e-6, 1e-5, 1e-4, 1e-3, 1e-2)


def gaps_of_greedy(W, opt, inp):
    a = [inp[k] for k in ('segs_feat', 'num', 'ppls', 'ppls_feat', 'sample_idx', 'pnt_mask')]
    with torch.no_grad():
        pre = O.preamble(W, opt, *a)
        B, H, L = a[0].shape[0], opt.rnn_size, opt.seq_length
        T, P = opt.num_sampled_frm, opt.num_prop_per_frm
        unk = int(opt.wtoi['UNK'])
        pm = a[5]
        state = (torch.zeros(2, B, H), torch.zeros(2, B, H))
        it = torch.zeros(B, dtype=torch.long)
        tok, reg = [], []
        for t in range(L):
            out, state, att2, _ = O.core_step(W, O.embed_word(W, it), pre, pm, pm, state)
            lp = O.word_logprobs(W, out)
            v, i = torch.topk(lp, 3, dim=1)
            keep = i[:, 0] != unk
            it = torch.where(keep, i[:, 0], i[:, 1])
            tok.append(torch.where(keep, v[:, 0] - v[:, 1], v[:, 1] - v[:, 2]))
            fr = att2.view(B, T, P)
            v2, _ = torch.topk(fr, 2, dim=2)
            live = v2[:, :, 0] > O.MIN_VALUE / 2                 # frame has at least one unmasked proposal
            single = live & (v2[:, :, 1] <= O.MIN_VALUE / 2)       # ... exactly one: no competitor
            g = (v2[:, :, 0] - v2[:, :, 1])[live & ~single]
            reg.append(g)
        return torch.stack(tok, 1).reshape(-1), torch.cat(reg)


def summary(g):
    g = g.double()
    q = torch.quantile(g, torch.tensor([0.0, 0.001, 0.01, 0.1, 0.5], dtype=torch.float64))
    return {'decisions': int(g.numel()), 'min_gap': float(g.min()),
            'quantiles': {k: float(v) for k, v in zip(('min', 'p0.1%', 'p1%', 'p10%', 'median'), q)},
            'count_below': {'%g' % th: int((g < th).sum()) for th in THRESH}}


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else 'greedy_b256_v5000_ft10_trained'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    Bt = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    Bd = int(sys.argv[4]) if len(sys.argv) > 4 else 64
    out = {'noise_floor': 'HIP vs oracle: log-probabilities within ~1.2e-5 at B=256 (bench parity.max_abs_logprob_diff), '
                          'attention logits within ~1e-5 absolute (tanh_fast 2.5e-7 x |alpha_net| 512-term sum + fp32 order)'}
    opt, sd, inp = cases.build_case(case)
    tok, reg = gaps_of_greedy(sd, opt, inp)
    out[case] = {'weights': 'synthetic trained_like profile (committed reference case)', 'token': summary(tok), 'region': summary(reg)}
    print(json.dumps(out[case]), file=sys.stderr, flush=True)
    # ---- weights after real optimisation steps
    opt = opts.default_opt(vocab_size=1000, t_attn_size=10)
    for k, v in cases.GRAD_WEIGHTS.items():
        setattr(opt, k, v)
    sd = synth.init_state_dict(opt, seed=31, profile='default')
    W = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v) for k, v in sd.items()}
    params = [v for v in W.values() if v.requires_grad]
    fine = [W[k] for k in W if W[k].requires_grad and ('ctx2pool_grd' in k or 'vis_embed' in k)]
    rest = [W[k] for k in W if W[k].requires_grad and not ('ctx2pool_grd' in k or 'vis_embed' in k)]
    optim = torch.optim.Adam([{'params': rest, 'lr': 5e-4}, {'params': fine, 'lr': 5e-5}], betas=(0.8, 0.999))
    hist = []
    for s in range(steps):
        b = synth.trim_to_batch(synth.make_inputs(opt, Bt, seed=1000 + s, train=True))
        optim.zero_grad(set_to_none=True)
        lm, a2, gl, cl, _ = O.forward_train(W, opt, *[b[k] for k in synth.FORWARD_ORDER])
        (lm + opt.w_att2 * a2 + opt.w_grd * gl + opt.w_cls * cl).backward()
        torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], 0.1)
        optim.step()
        hist.append(float(lm))
        print('step %d lm %.4f' % (s, hist[-1]), file=sys.stderr, flush=True)
    Wt = {k: v.detach() for k, v in W.items()}
    inp = synth.make_inputs(opt, Bd, seed=77, train=False)
    tok, reg = gaps_of_greedy(Wt, opt, inp)
    out['after_%d_adam_steps' % steps] = {
        'weights': 'torch default init + %d steps of clip 0.1 + Adam 5e-4 (x0.1 fc7 / vis_embed) on fresh synthetic batches of %d '
                   "segments, oracle 'MLE' autograd, V=1000" % (steps, Bt),
        'lm_loss_first_last': [hist[0], hist[-1]], 'decode_batch': Bd, 'token': summary(tok), 'region': summary(reg)}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
