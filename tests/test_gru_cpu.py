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
