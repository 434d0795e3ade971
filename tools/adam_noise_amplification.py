"""How far do two fp32 implementations of main.train drift apart over a few optimisation steps when their gradients differ by
the rounding-level direction error measured between the HIP path and the reference (~1e-3 of a parameter's gradient norm,
DESIGN.md section 5)?  The CPU oracle runs the four-step trajectory case twice: exactly, and with every parameter gradient
perturbed by Gaussian noise of relative Frobenius size EPS before clip + Adam.  Adam's first updates are lr * m / sqrt(v) =
+-lr per element whatever the gradient's size, so elements whose gradient is below the noise flip their update.
    python tools/adam_noise_amplification.py [eps]     ->  per-step |loss difference| and relative gradient-norm difference"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import gvd_amd  # noqa: E402,F401
from gvd_amd import synth  # noqa: E402
from oracle import cases, gvd_oracle as O  # noqa: E402

EPS = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3
name = 'traj4_b4_v1000_ft10_trained'
torch.set_num_threads(6)


def run(eps, seed=0):
    opt, sd, _ = cases.build_case(name)
    W = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v) for k, v in sd.items()}
    groups = [{'params': [v], 'lr': 5e-4 * (0.1 if ('ctx2pool_grd' in k or 'vis_embed' in k) else 1.0)}
              for k, v in W.items() if torch.is_tensor(v) and v.requires_grad]
    optim = torch.optim.Adam(groups)
    w = cases.GRAD_WEIGHTS
    g = torch.Generator().manual_seed(seed)
    out = []
    for batch in cases.traj_batches(name):
        optim.zero_grad(set_to_none=True)
        lm, a2, gl, cl, _ = O.forward_train(W, opt, *[batch[k] for k in synth.FORWARD_ORDER])
        (lm + w['w_att2'] * a2 + w['w_grd'] * gl + w['w_cls'] * cl).backward()
        have = [v for v in W.values() if torch.is_tensor(v) and v.requires_grad and v.grad is not None]
        if eps > 0:
            for v in have:
                n = torch.randn(v.grad.shape, generator=g)
                v.grad.add_(n * (eps * float(v.grad.norm()) / max(float(n.norm()), 1e-30)))
        total = float(torch.nn.utils.clip_grad_norm_(have, 0.1))
        optim.step()
        out.append(([float(lm), float(a2), float(gl), float(cl)], total))
    return out


a, b = run(0.0), run(EPS)
for i, ((la, na), (lb, nb)) in enumerate(zip(a, b)):
    print('step %d: |loss difference| %s   gradient norm %.5f vs %.5f (%.2e relative)'
          % (i + 1, ['%.2e' % abs(x - y) for x, y in zip(la, lb)], na, nb, abs(na - nb) / na), flush=True)
