"""The hand-scheduled GRU BPTT (gru_fn.GruLayerFn) against autograd through torch.nn.GRU, on CPU with the torch
stand-in kernels (tests/torch_backend.py): validates the backward algebra / orchestration; the HIP kernels themselves are
validated on the GPU (tests/test_gpu_train.py)."""
import pytest
import torch
import torch.nn as nn

import gvd_amd  # noqa: F401
from gvd_amd import gru_fn
from tests import torch_backend


@pytest.mark.parametrize('B,T', [(3, 5), (2, 1), (4, 2)])
def test_gru_bptt_matches_autograd_through_nn_gru(monkeypatch, B, T):
    monkeypatch.setattr(gru_fn, 'K', torch_backend)
    torch.manual_seed(5)
    In, Hh = 12, 6
    ref = nn.GRU(In, Hh, 2, dropout=0.0, bidirectional=True, batch_first=True).double()
    mine = nn.GRU(In, Hh, 2, dropout=0.0, bidirectional=True, batch_first=True).double()
    mine.load_state_dict(ref.state_dict())
    x = torch.randn(B, T, In, dtype=torch.float64)
    G = torch.randn(B, T, 2 * Hh, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)[0]
    (yr * G).sum().backward()
    xm = x.clone().requires_grad_(True)
    ym = gru_fn.gru_bidir_2layer_train(xm, mine)
    assert torch.allclose(ym, yr, atol=1e-12)
    (ym * G).sum().backward()
    assert torch.allclose(xm.grad, xr.grad, rtol=1e-9, atol=1e-12)
    for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        assert torch.allclose(p.grad, q.grad, rtol=1e-9, atol=1e-12), n


@pytest.mark.parametrize('B,T', [(3, 5), (2, 1), (4, 2)])
def test_lstm_bptt_matches_autograd_through_nn_lstm(monkeypatch, B, T):
    """`--t_attn_mode bilstm` (model.py:145-149): lstm_fn.LstmLayerFn (the hand-scheduled BPTT over the saved gates / cell
    states) against autograd through torch.nn.LSTM, fp64, stand-in kernels."""
    from gvd_amd import lstm_fn
    monkeypatch.setattr(gru_fn, 'K', torch_backend)
    monkeypatch.setattr(lstm_fn, 'K', torch_backend)
    torch.manual_seed(6)
    In, Hh = 12, 6
    ref = nn.LSTM(In, Hh, 2, dropout=0.0, bidirectional=True, batch_first=True).double()
    mine = nn.LSTM(In, Hh, 2, dropout=0.0, bidirectional=True, batch_first=True).double()
    mine.load_state_dict(ref.state_dict())
    x = torch.randn(B, T, In, dtype=torch.float64)
    G = torch.randn(B, T, 2 * Hh, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)[0]
    (yr * G).sum().backward()
    xm = x.clone().requires_grad_(True)
    ym = lstm_fn.lstm_bidir_2layer_train(xm, mine)
    assert torch.allclose(ym, yr, atol=1e-12)
    (ym * G).sum().backward()
    assert torch.allclose(xm.grad, xr.grad, rtol=1e-9, atol=1e-12)
    for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        assert torch.allclose(p.grad, q.grad, rtol=1e-9, atol=1e-12), n


def test_oracle_lstm_forms_agree_with_nn_lstm():
    """oracle.gvd_oracle.lstm_bidir_2layer(_loop): the explicit time loop, the fused library call and torch.nn.LSTM itself."""
    from oracle import gvd_oracle as O
    torch.manual_seed(7)
    ref = nn.LSTM(16, 8, 2, dropout=0.0, bidirectional=True, batch_first=True)
    W = {'context_enc.' + n: p.detach() for n, p in ref.named_parameters()}
    x = torch.randn(3, 7, 16)
    want = ref(x)[0].detach()
    assert torch.allclose(O.lstm_bidir_2layer_loop(x, W), want, atol=1e-6)
    assert torch.allclose(O.lstm_bidir_2layer(x, W), want, atol=1e-6)
