"""Which non-README option values of the reference's opts surface does the REFERENCE ITSELF run?  Build container only
(imports /root/reference through oracle/ref_harness.py).  For every value of region_attn_mode / att_input_mode /
transfer_mode / t_attn_mode (opts.py:58-63) outside the README recipe: construct misc.AttModel.TopDownModel with the README
dimensions, run one greedy 'sample' and one 'MLE' forward on a 2-segment synthetic batch, record pass / the exception.
Output: profiles/r05/reference_option_survey.json (the evidence behind DESIGN.md section 8's rejected-option table)."""
import json
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gvd_amd  # noqa: E402,F401
from gvd_amd import opts, synth  # noqa: E402
from oracle import ref_harness  # noqa: E402

CASES = [('region_attn_mode', v) for v in ('mix', 'add', 'cat', 'mix_mul', 'dp')] + \
        [('att_input_mode', v) for v in ('featmap', 'region', 'dual_region')] + \
        [('transfer_mode', v) for v in ('none', 'glove', 'both')] + [('t_attn_mode', 'bilstm')]


def run(key, val):
    opt = opts.default_opt(vocab_size=300, t_attn_size=10, **{key: val})
    out = {'option': key, 'value': val}
    try:
        ref, _ = ref_harness.construct_reference_fresh(opt, 0)
        ref.eval()
        out['construct'] = 'ok'
    except Exception as e:          # noqa: BLE001
        out['construct'] = '%s: %s' % (type(e).__name__, str(e)[:200])
        return out
    inp = synth.trim_to_batch(synth.make_inputs(opt, 2, seed=0, train=True))
    for mode in ('sample', 'MLE'):
        try:
            with torch.no_grad():
                if mode == 'sample':
                    ref._sample(inp['segs_feat'], inp['ppls'], inp['num'], inp['ppls_feat'], inp['sample_idx'], inp['pnt_mask'],
                                {'sample_max': 1, 'beam_size': 1})
                else:
                    ref(*synth.as_args(inp), 'MLE')
            out[mode] = 'ok'
        except Exception as e:      # noqa: BLE001
            tb = traceback.extract_tb(e.__traceback__)
            where = [f for f in tb if '/root/reference' in f.filename]
            out[mode] = '%s: %s%s' % (type(e).__name__, str(e)[:160],
                                      (' @ %s:%d' % (os.path.relpath(where[-1].filename, '/root/reference'), where[-1].lineno)) if where else '')
    return out


if __name__ == '__main__':
    res = [run(k, v) for k, v in CASES]
    os.makedirs('profiles/r05', exist_ok=True)
    with open('profiles/r05/reference_option_survey.json', 'w') as f:
        json.dump(res, f, indent=1)
    for r in res:
        print(json.dumps(r))
