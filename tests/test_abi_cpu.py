"""CPU: the C-ABI library loads and exports every function include/gvd_hip.h declares (no compute calls),
and the host side refuses to run the hot path without the GPU (no silent fallback)."""
import ctypes
import os
import re

import pytest
import torch

import gvd_amd
from gvd_amd import att_model, build, hip, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'gvd_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(gvd_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    path = build.library_path()
    assert os.path.exists(path), 'run __graft_entry__.build() first'
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), 'libgvd_hip.so does not export %s' % n
    # the Python binding table covers exactly the declared functions
    assert sorted(hip.EXPORTS) == names
    assert b'gfx950' in hip.lib().gvd_version()


def test_no_cpu_fallback():
    opt = gvd_amd.opts.default_opt(vocab_size=50, t_attn_size=4)
    m = att_model.TopDownModel(opt)
    inp = gvd_amd.synth.make_inputs(opt, 1, seed=0, train=False)
    with pytest.raises(hip.GvdHipError):
        m(*gvd_amd.synth.as_args(inp), 'sample', {})
    with pytest.raises(hip.GvdHipError):
        ops.gemm_nt(torch.zeros(2, 32), torch.zeros(4, 32))


def test_unsupported_configuration_is_rejected():
    with pytest.raises(NotImplementedError):
        att_model.TopDownModel(gvd_amd.opts.default_opt(vocab_size=50, att_model='transformer'))


def test_integration_doc_stub_matches_the_binding():
    """The ctypes stub INTEGRATION.md shows a maintainer (gvd_attn_side) lists the fields of the real binding, in order."""
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    body = doc[doc.index('class AttnSide(C.Structure)'):]
    body = body[:body.index('\n\n')]
    names = re.findall(r"\('([A-Za-z_0-9]+)',", body)
    assert names == [f[0] for f in hip.AttnSide._fields_], names
