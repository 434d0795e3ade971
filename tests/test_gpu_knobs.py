"""-m gpu: every non-default value of the GVD_* A/B knobs still reproduces the reference goldens.  Several knobs
are read once per process, so each setting runs tools/knob_check.py in its own process; knobs that act on different
components are grouped into one process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SETTINGS = [
    ('decode', dict(GVD_PERSISTENT='0', GVD_ATTN_NT='1', GVD_ATTN_CHUNK='32', GVD_GEMV_KS='2', GVD_GEMM_SMALL='0',
                    GVD_FC7_ROWMAP='0', GVD_GRU_BARRIER='cg', GVD_FLASH_SKEW='0', GVD_COOP_LAUNCH='1')),
    ('decode', dict(GVD_ATTN_NT='0', GVD_GEMV_KS='1', GVD_GEMM_VARIANT='1', GVD_GEMM_BIG='128', GVD_SIDE_FUSED='0', GVD_GRU_HU='16')),
    ('decode', dict(GVD_COMPACT='0', GVD_POOL_EMBED_OWN='0', GVD_GEMM_VARIANT='0')),
    ('decode', dict(GVD_COMPACT='0', GVD_ENC_FUSED='0', GVD_GEMM_VARIANT='2')),               # first flash kernels (16x16x4)
    ('decode', dict(GVD_ENC_FUSED='0', GVD_FLASH_V16='0', GVD_FLASH_GLDS='1')),               # 32x32x2 flash kernel, LDS-DMA staging
    ('decode', dict(GVD_ENC_FUSED='0', GVD_FLASH='0')),                                       # library attention chain
    ('beam', dict(GVD_ATTN_GROUPED='0', GVD_BEAM_FUSED='0')),
    ('train', dict(GVD_ENC_TRAIN_MFMA='0', GVD_LN_FUSED_BWD='0', GVD_P5_FUSED_TRAIN='0', GVD_GRU_TRAIN='0')),
    ('train', dict(GVD_ENC_HEADS_MERGED='0', GVD_TRAIN_HEAD_PAD='192')),
    ('train', dict(GVD_TRAIN_FUSED_ELEMENTWISE='0')),
]


@pytest.mark.parametrize('mode,env', SETTINGS, ids=['%s-%s' % (m, '+'.join('%s=%s' % (k[4:], v) for k, v in e.items()))
                                                     for m, e in SETTINGS])
def test_non_default_knobs_reproduce_the_goldens(mode, env):
    e = {k: v for k, v in os.environ.items() if not k.startswith('GVD_')}
    e.update(env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'knob_check.py'), mode], env=e, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-1500:])
