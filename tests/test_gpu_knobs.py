"""-m gpu: every non-default value of the runtime knobs that select a code path still reproduces
the reference goldens, and the persistent kernels' timeout path recomputes instead of failing.  The knobs are read once
per process (or are process-wide state), so each setting runs tools/knob_check.py in its own process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SETTINGS = [
    # kernel-per-op decoder instead of the persistent one, library grid sync + cooperative launch for the bi-GRU
    ('decode', dict(GVD_PERSISTENT='0', GVD_GRU_BARRIER='cg')),
    ('decode', dict(GVD_COOP_LAUNCH='1')),
    ('decode', dict(GVD_COMPACT='0')),                       # dense preamble (every masked proposal computed)
    ('train', dict(GVD_TRAIN_COMPACT='1')),                  # compacted training layout
    ('train', dict(GVD_GRU_BARRIER='cg')),
    # forced grid-barrier timeouts (spin limit 1): forward / Trainer.step switch the persistent kernels off, recompute and
    # still deliver the reference result (tools/knob_check.py asserts the warning and the switch)
    ('timeout_decode', dict(GVD_SPIN_LIMIT='1')),
    ('timeout_train', dict(GVD_SPIN_LIMIT='1')),
]


@pytest.mark.parametrize('mode,env', SETTINGS, ids=['%s-%s' % (m, '+'.join('%s=%s' % (k[4:], v) for k, v in e.items()))
                                                     for m, e in SETTINGS])
def test_non_default_knobs_reproduce_the_goldens(mode, env):
    e = {k: v for k, v in os.environ.items() if not k.startswith('GVD_')}
    e.update(env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'knob_check.py'), mode], env=e, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-1500:])
