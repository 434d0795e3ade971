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


def test_struct_layouts_match_the_header(tmp_path):
    """Every struct of include/gvd_hip.h against its ctypes mirror in hip.py: total size and the offset of every field,
    measured by a C program compiled from the header itself (gcc; the header is plain C)."""
    import subprocess
    pairs = {'gvd_gemm_seg': hip.GemmSeg, 'gvd_gemm_args': hip.GemmArgs, 'gvd_lstm_args': hip.LstmArgs,
             'gvd_attn_side': hip.AttnSide, 'gvd_beam_step_args': hip.BeamStepArgs, 'gvd_greedy_args': hip.GreedyArgs,
             'gvd_opt_group': hip.OptGroup, 'gvd_dx_group': hip.DxGroup}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "gvd_hip.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append('  printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    got = {}
    for ln in subprocess.check_output([str(exe)]).decode().split('\n'):
        if ln.strip():
            a, b, c = ln.split()
            got[(a, b)] = int(c)
    for cname, cls in pairs.items():
        assert got[(cname, 'size')] == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)
    assert hip.OPT_MAX_TENSORS == 32 and hip.OPT_CHUNK == hip.lib().gvd_opt_chunk()


def test_optimizer_launch_packing():
    """optim.ClipAdam._launches: tensors split into launches of <= 32, chunk prefix sums, running partial offsets."""
    from gvd_amd.optim import ClipAdam
    sizes = [5, hip.OPT_CHUNK, hip.OPT_CHUNK + 1, 3 * hip.OPT_CHUNK] * 10          # 40 tensors -> 2 launches
    items = [{'g': torch.zeros(n), 'n': n} for n in sizes]
    groups = list(ClipAdam._launches(items))
    assert [g.count for g in groups] == [32, 8]
    chunks = [-(-n // hip.OPT_CHUNK) for n in sizes]
    assert groups[0].part0 == 0 and groups[1].part0 == sum(chunks[:32])
    for gi, g in enumerate(groups):
        part = chunks[32 * gi:32 * gi + g.count]
        assert list(g.chunk0[:g.count + 1]) == [sum(part[:i]) for i in range(g.count + 1)]
        assert list(g.n[:g.count]) == sizes[32 * gi:32 * gi + g.count]


def test_function_signatures_match_the_header():
    """Every prototype of include/gvd_hip.h against the ctypes signature in hip._SIG: parameter count and, per parameter,
    the kind (pointer / int / int64 / uint64 / float / size_t) - a swapped or missing argument in the binding would
    otherwise only show up on the GPU box."""
    src = open(os.path.join(ROOT, 'include', 'gvd_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    protos = re.findall(r'\n\s*([A-Za-z_][A-Za-z_0-9 \*]*?)\s*\b(gvd_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;', src)
    assert len(protos) >= 50

    def kind(ctype):
        t = ctype.strip()
        if t.endswith('*') or t in ('gvd_stream_t',):
            return 'ptr'
        return {'int': 'int', 'int64_t': 'i64', 'uint64_t': 'u64', 'float': 'float', 'size_t': 'u64', 'void': 'void',
                'const char*': 'ptr'}[t.replace('const ', '')]       # (size_t and uint64_t are one ctypes class here)

    def ckind(ct):
        if ct is None:
            return 'void'
        if ct in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(ct, 'contents') or (isinstance(ct, type) and issubclass(ct, ctypes._Pointer)):
            return 'ptr'
        return {ctypes.c_int: 'int', ctypes.c_int64: 'i64', ctypes.c_uint64: 'u64', ctypes.c_float: 'float',
                ctypes.c_size_t: 'u64'}[ct]
    seen = set()
    for ret, name, params in protos:
        seen.add(name)
        res, args = hip._SIG[name]
        plist = [p.strip() for p in params.split(',')] if params.strip() not in ('', 'void') else []
        want = []
        for p in plist:
            m = re.match(r'^(.*?)([A-Za-z_][A-Za-z_0-9]*)$', p)          # type, then the parameter name
            want.append(kind(m.group(1)))
        got = [ckind(a) for a in args]
        assert got == want, '%s: binding %s vs header %s' % (name, got, want)
        rk = kind(ret)
        # (c_size_t and c_int64 results are declared as such; a `const char*` result is a pointer)
        assert ckind(res) == rk, '%s: result %s vs header %s' % (name, ckind(res), rk)
    assert seen == set(hip.EXPORTS)
