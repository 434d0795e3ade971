"""CPU: control flow of train.Trainer.step around the status word of a step, with a stand-in model (the HIP model needs the
GPU): a loader-contract violation met on the compacted training layout makes the trainer switch to the full row set and run
the step again (compute, don't raise); without the compaction - or a second time - it raises; a kernel error always raises
and never reaches the optimiser."""
import argparse

import pytest
import torch
import torch.nn as nn

from gvd_amd import train
from gvd_amd.hip import GvdHipError


class _Fake(nn.Module):
    def __init__(self, script):
        super().__init__()
        self.w = nn.Parameter(torch.ones(3))
        self.ctx2pool_grd = nn.Linear(2, 2)          # (a parameter of the x0.1 learning-rate group, main.py:660-677)
        self.script = list(script)                   # per forward: (bad, contract)
        self.calls = 0

    def forward(self, *args):
        self.calls += 1
        loss = (self.w ** 2).sum() + self.ctx2pool_grd.weight.sum() * 0
        return loss.view(1), loss.view(1) * 0, loss.view(1) * 0, loss.view(1) * 0

    def kernel_status_counts(self):
        bad, contract = self.script.pop(0)
        return torch.tensor([bad, contract], dtype=torch.int64)

    @staticmethod
    def raise_for_status(bad, contract):
        raise GvdHipError('bad=%d contract=%d' % (bad, contract))


def _opt():
    return argparse.Namespace(learning_rate=0.1, weight_decay=0, optim_alpha=0.9, optim_beta=0.999, optim='adam',
                              grad_clip=0.1, w_att2=0.05, w_grd=0.0, w_cls=0.1)


def test_contract_violation_on_the_compacted_layout_reruns_the_step_on_all_rows(monkeypatch):
    monkeypatch.setenv('GVD_TRAIN_COMPACT', '1')
    m = _Fake([(0, 2), (0, 0), (0, 0)])
    tr = train.Trainer(m, _opt())
    before = m.w.detach().clone()
    tr.step(())
    assert m.calls == 2 and m._train_compact_off is True          # one retry, compaction off for good
    assert not torch.equal(m.w.detach(), before)                  # ... and the retried step reached the optimiser
    tr.step(())
    assert m.calls == 3


def test_contract_violation_without_compaction_raises_before_the_optimiser(monkeypatch):
    monkeypatch.delenv('GVD_TRAIN_COMPACT', raising=False)
    m = _Fake([(0, 1)])
    tr = train.Trainer(m, _opt())
    before = m.w.detach().clone()
    with pytest.raises(GvdHipError):
        tr.step(())
    assert m.calls == 1 and torch.equal(m.w.detach(), before)
    # compaction already switched off for this model: a further violation raises too
    monkeypatch.setenv('GVD_TRAIN_COMPACT', '1')
    m2 = _Fake([(0, 1), (0, 1)])
    tr2 = train.Trainer(m2, _opt())
    with pytest.raises(GvdHipError):
        tr2.step(())
    assert m2.calls == 2 and m2._train_compact_off is True


def test_kernel_error_raises_and_never_retries(monkeypatch):
    monkeypatch.setenv('GVD_TRAIN_COMPACT', '1')
    m = _Fake([(1, 1)])
    tr = train.Trainer(m, _opt())
    before = m.w.detach().clone()
    with pytest.raises(GvdHipError):
        tr.step(())
    assert m.calls == 1 and torch.equal(m.w.detach(), before) and not getattr(m, '_train_compact_off', False)
