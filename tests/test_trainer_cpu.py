"""CPU: control flow of train.Trainer.step around the status word of a step, with a stand-in model (the HIP model needs the
GPU): a loader-contract violation met on the compacted training layout makes the trainer switch to the full row set and run
the step again (compute, don't raise); without the compaction - or a second time - it raises; a grid-barrier timeout of a
persistent kernel switches those kernels off for the process and runs the step again, a second one raises; an invalid
attempt never reaches the optimiser; status words left over from earlier inference calls are dropped, not judged."""
import argparse

import pytest
import torch
import torch.nn as nn

from gvd_amd import ops, train
from gvd_amd.hip import GvdHipError


@pytest.fixture(autouse=True)
def _persistent_kernels_on():
    ops._persistent.update(on=True, timeouts=0)
    yield
    ops._persistent.update(on=True, timeouts=0)


class _Fake(nn.Module):
    def __init__(self, script):
        super().__init__()
        self.w = nn.Parameter(torch.ones(3))
        self.ctx2pool_grd = nn.Linear(2, 2)          # (a parameter of the x0.1 learning-rate group, main.py:660-677)
        self.script = list(script)                   # per forward: (bad, contract)
        self.calls = 0
        self.stale = None                            # status words of launches BEFORE the step (an unchecked _sample)
        self.fresh = False

    def forward(self, *args):
        self.calls += 1
        self.fresh = True
        loss = (self.w ** 2).sum() + self.ctx2pool_grd.weight.sum() * 0
        return loss.view(1), loss.view(1) * 0, loss.view(1) * 0, loss.view(1) * 0

    def kernel_status_counts(self):
        if not self.fresh:                           # nothing launched since the last call (or only the stale words)
            stale, self.stale = self.stale, None
            return None if stale is None else torch.tensor(stale, dtype=torch.int64)
        self.fresh = False
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


def test_barrier_timeout_switches_the_persistent_kernels_off_and_reruns_the_step(monkeypatch):
    monkeypatch.delenv('GVD_TRAIN_COMPACT', raising=False)
    m = _Fake([(2, 0), (0, 0), (1, 0), (1, 0)])
    tr = train.Trainer(m, _opt())
    before = m.w.detach().clone()
    with pytest.warns(RuntimeWarning, match='grid-barrier timeout'):
        tr.step(())
    assert m.calls == 2 and not ops.persistent_kernels_enabled() and ops._persistent['timeouts'] == 2
    assert not torch.equal(m.w.detach(), before)                  # the retried step reached the optimiser
    # a later step whose ONE retry (decided from the status word alone - the same on every rank - not from this process's
    # switch, which is already off here) is invalid too is a real kernel error: raises, no update
    after = m.w.detach().clone()
    with pytest.raises(GvdHipError):
        tr.step(())
    assert m.calls == 4 and torch.equal(m.w.detach(), after)


def test_retry_decision_does_not_depend_on_the_process_local_switch(monkeypatch):
    """Data-parallel hang of ADVICE r4: rank A switched its persistent kernels off earlier (a rank-local validation pass
    timed out), rank B did not; a later training step times out on B -> the MAX-reduced word says bad on BOTH.  Both must
    take the same branch (retry), whatever their local switch says."""
    monkeypatch.delenv('GVD_TRAIN_COMPACT', raising=False)
    calls = []
    for switch_on in (True, False):                  # rank B, rank A
        ops._persistent.update(on=switch_on, timeouts=0)
        m = _Fake([(1, 0), (0, 0)])
        tr = train.Trainer(m, _opt())
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            tr.step(())
        calls.append(m.calls)
    assert calls == [2, 2]


def test_timeout_and_contract_violation_in_one_step_take_one_retry(monkeypatch):
    monkeypatch.setenv('GVD_TRAIN_COMPACT', '1')
    m = _Fake([(1, 1), (0, 0)])
    tr = train.Trainer(m, _opt())
    with pytest.warns(RuntimeWarning):
        tr.step(())
    assert m.calls == 2 and m._train_compact_off is True and not ops.persistent_kernels_enabled()


def test_status_words_left_by_earlier_inference_calls_do_not_fail_the_step(monkeypatch):
    monkeypatch.delenv('GVD_TRAIN_COMPACT', raising=False)
    m = _Fake([(0, 0)])
    m.stale = (0, 3)                                 # e.g. a direct _sample() on inputs that break the zero-row contract
    tr = train.Trainer(m, _opt())
    before = m.w.detach().clone()
    tr.step(())
    assert m.calls == 1 and not torch.equal(m.w.detach(), before)
