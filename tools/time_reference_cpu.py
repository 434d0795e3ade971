"""Time the REAL reference (misc.AttModel.TopDownModel imported from /root/reference through oracle/ref_harness.py) on this
container's host cores and write profiles/cpu_reference_timing.json.  Build container only (the GPU box has no
/root/reference); bench.py attaches the committed file to its `cpu_baseline` block so the reference's own CPU rate is shown
beside the oracle's rate measured on the GPU box.

    python tools/time_reference_cpu.py            # BASELINE configs[0]: greedy B=4, L=20, 10x100 regions, Ft=10, V=5000
"""
import json
import os
import platform
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gvd_amd  # noqa: E402
from gvd_amd import opts, synth  # noqa: E402
from oracle import gvd_oracle as O, ref_harness  # noqa: E402


def rate(fn, B, seconds=20.0):
    fn()
    n, t0 = 0, time.time()
    while time.time() - t0 < seconds:
        fn()
        n += 1
    dt = time.time() - t0
    return round(B * n / dt, 3), n, round(dt, 1)


def main():
    ncpu = os.cpu_count()
    torch.set_num_threads(ncpu)
    opt = opts.default_opt(vocab_size=5000, t_attn_size=10)
    sd = synth.init_state_dict(opt, seed=0, profile='trained_like')
    ref = ref_harness.build_reference_model(opt, sd).eval()
    out = {'host': platform.processor() or platform.machine(), 'cores': ncpu, 'torch': torch.__version__,
           'note': 'REAL reference code from /root/reference (eval mode, no CUDA) on the build container; the oracle '
                   'port is timed beside it on the same cores'}
    with torch.no_grad():
        for B in (4, 32):
            inp = synth.make_inputs(opt, B, seed=0, train=False)
            a = [inp[k] for k in ('segs_feat', 'num', 'ppls', 'ppls_feat', 'sample_idx', 'pnt_mask')]
            r, n, dt = rate(lambda: ref._sample(inp['segs_feat'], inp['ppls'], inp['num'], inp['ppls_feat'],
                                                inp['sample_idx'], inp['pnt_mask'], {'sample_max': 1, 'beam_size': 1}), B)
            o, n2, dt2 = rate(lambda: O.sample_greedy(sd, opt, *a), B)
            out['greedy_b%d' % B] = {'reference_captions_per_s': r, 'reference_calls': n, 'reference_seconds': dt,
                                     'oracle_port_captions_per_s': o, 'oracle_calls': n2}
            print(B, out['greedy_b%d' % B])
    os.makedirs(os.path.join(ROOT, 'profiles'), exist_ok=True)
    with open(os.path.join(ROOT, 'profiles', 'cpu_reference_timing.json'), 'w') as f:
        json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
