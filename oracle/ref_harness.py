"""Harness that imports the REAL reference model from /root/reference (build container only).

TEST INFRASTRUCTURE — not part of the product path.  Nothing here is importable on the GPU box
(/root/reference does not exist there); it is used only by oracle/make_golden.py and by the CPU tests
that pin oracle/gvd_oracle.py against the reference (tests/test_oracle_vs_reference.py, skipped when the
reference tree is absent).

No reference file is edited or copied.  The harness only prepares the process the way SURVEY.md
Appendix B describes:
  1. CWD holds synthetic data/detectron_weights/*.pkl (model.py:173-185 reads them relative to CWD);
  2. Tensor.masked_fill_/masked_fill/masked_select accept uint8 masks again (PyTorch-1.1 semantics the
     reference was written for, README.md:46,52);
  3. nn.Dropout(inplace=True) -> inplace=False when gradients are needed (torch-2 autograd);
  4. our synthetic state_dict (synth.init_state_dict) is loaded with strict=True, which also proves
     the parameter names/shapes of the boundary.
"""
import contextlib
import io
import os
import pickle
import sys
import tempfile

import numpy as np
import torch

REFERENCE_ROOT = '/root/reference'


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'misc'))


_shimmed = False


def _install_uint8_mask_shim():
    global _shimmed
    if _shimmed:
        return
    for name in ('masked_fill_', 'masked_fill', 'masked_select'):
        orig = getattr(torch.Tensor, name)

        def wrap(o):
            def f(self, m, *a, **k):
                return o(self, m.bool() if m.dtype == torch.uint8 else m, *a, **k)
            return f
        setattr(torch.Tensor, name, wrap(orig))
    _ms = torch.masked_select
    torch.masked_select = lambda x, m, **k: _ms(x, m.bool() if m.dtype == torch.uint8 else m, **k)
    _shimmed = True


_workdir = None


def _enter_workdir():
    """chdir into a temp dir holding synthetic Detectron pickles (shapes from model.py:173-185)."""
    global _workdir
    if _workdir is None:
        _workdir = tempfile.mkdtemp(prefix='gvd_ref_')
        d = os.path.join(_workdir, 'data', 'detectron_weights')
        os.makedirs(d)
        rng = np.random.RandomState(0)
        for n, s in (('fc7_w', (2048, 2048)), ('fc7_b', (2048,)),
                     ('cls_score_w', (1601, 2048)), ('cls_score_b', (1601,))):
            with open(os.path.join(d, n + '.pkl'), 'wb') as f:
                pickle.dump((0.01 * rng.randn(*s)).astype(np.float32), f)
    os.chdir(_workdir)


def build_reference_model(opt, state_dict, need_grad=False):
    """Instantiate misc.AttModel.TopDownModel(opt) from /root/reference and load `state_dict`."""
    assert reference_available(), 'reference tree not present (GPU box?)'
    _install_uint8_mask_shim()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    cwd = os.getcwd()
    _enter_workdir()
    try:
        from misc import AttModel  # noqa: the reference module
        with contextlib.redirect_stdout(io.StringIO()):   # silences the NN-class match prints (model.py:204)
            model = AttModel.TopDownModel(opt)
    finally:
        os.chdir(cwd)
    missing = model.load_state_dict(state_dict, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    if need_grad:
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.inplace = False
    return model


@contextlib.contextmanager
def beam_shim(model):
    """Run-time repair that lets the reference's OWN `CaptionModel.beam_search` / `_sample_beam`
    (CaptionModelBU.py:24-185, model.py:627-742) execute on CPU, so that beam parity is pinned on the reference's
    code instead of on a re-reading of it.  Nothing in /root/reference is edited:
      * CaptionModelBU.py:179-181 calls `self.core(...)` with 12 positional arguments (a stray all-zero tensor before
        `beam_sim_mat_static` and a trailing `self`) while TopDownCore.forward takes 10 (AttModel.py:134) -> the
        instance's `core.forward` is wrapped to drop exactly those two;
      * `.cuda()` is hard-coded at CaptionModelBU.py:136 and model.py:738-740 -> identity while the shim is active.
    """
    core = model.core
    bound = core.forward

    def forward(*a):
        if len(a) == 12:
            a = a[:9] + (a[10],)
        return bound(*a)
    core.forward = forward
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        yield model
    finally:
        torch.Tensor.cuda = orig_cuda
        del core.forward


def reference_beam_sample(model, inp, beam_size):
    """The reference's beam `'sample'` under `beam_shim` -> (seq i64[B,L], seqLogprobs f32[B,L], att2 i64[B,L])."""
    with beam_shim(model), torch.no_grad():
        return model._sample(inp['segs_feat'], inp['ppls'], inp['num'], inp['ppls_feat'], inp['sample_idx'],
                             inp['pnt_mask'], {'sample_max': 1, 'beam_size': beam_size})


def construct_reference_fresh(opt, seed):
    """`torch.manual_seed(seed); misc.AttModel.TopDownModel(opt)` in the harness work directory (synthetic Detectron
    pickles) WITHOUT loading a state_dict: the reference's own initialisation incl. its knowledge transfer
    (model.py:173-216).  Returns (model, workdir) so the caller can construct its own model in the same CWD."""
    assert reference_available()
    _install_uint8_mask_shim()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    cwd = os.getcwd()
    _enter_workdir()
    try:
        from misc import AttModel  # noqa
        torch.manual_seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            model = AttModel.TopDownModel(opt)
    finally:
        os.chdir(cwd)
    return model, _workdir
