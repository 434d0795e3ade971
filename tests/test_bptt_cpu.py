"""The hand-written BPTT (decoder_bwd.DecoderLoopFn) against autograd through the oracle's step
(oracle core_step == reference TopDownCore.forward), on CPU with the torch stand-in kernels
(tests/torch_backend.py).  This validates the backward ALGEBRA and orchestration; the HIP kernels
themselves are validated on the GPU (tests/test_gpu_train.py)."""
import pytest
import torch

import gvd_amd
from gvd_amd import decoder_bwd, decoder_fn
from oracle import gvd_oracle as O
from tests import torch_backend


@pytest.fixture
def fake_kernels(monkeypatch):
    monkeypatch.setattr(decoder_fn, 'K', torch_backend)


def _problem(seed, B=3, H=16, A=8, E=8, R=7, Ft=4, Lc=5, per_step=True):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    W = {'core.att_lstm.weight_ih': r(4 * H, E + H) * 0.3, 'core.att_lstm.weight_hh': r(4 * H, H) * 0.3,
         'core.att_lstm.bias_ih': r(4 * H) * 0.1, 'core.att_lstm.bias_hh': r(4 * H) * 0.1,
         'core.lang_lstm.weight_ih': r(4 * H, 2 * H) * 0.3, 'core.lang_lstm.weight_hh': r(4 * H, H) * 0.3,
         'core.lang_lstm.bias_ih': r(4 * H) * 0.1, 'core.lang_lstm.bias_hh': r(4 * H) * 0.1,
         'core.attention.h2att.weight': r(A, H) * 0.3, 'core.attention.h2att.bias': r(A) * 0.1,
         'core.attention.alpha_net.weight': r(1, A), 'core.attention.alpha_net.bias': r(1) * 0.1,
         'core.attention2.h2att.weight': r(A, H) * 0.3, 'core.attention2.h2att.bias': r(A) * 0.1,
         'core.attention2.alpha_net.weight': r(1, A), 'core.attention2.alpha_net.bias': r(1) * 0.1}
    pre = dict(fc=r(B, H), conv=r(B, Ft, H), p_conv=r(B, Ft, A), pool=r(B, R, H), p_pool=r(B, R, A))
    xt_all = r(B, Lc, E)
    att_mask = (torch.rand(B, R + 1, generator=g) < 0.25).to(torch.uint8)
    att_mask[0, 1:] = 1                                    # one fully masked sample
    if per_step:
        pnt_masks = ((torch.rand(B, Lc, R + 1, generator=g) < 0.4).to(torch.uint8) | att_mask.unsqueeze(1))
    else:
        pnt_masks = att_mask.clone()
    Gh, Ga = r(B, Lc, H), r(B, Lc, R)
    return W, pre, xt_all, att_mask, pnt_masks, Gh, Ga


KEYMAP = dict(att_w_ih='core.att_lstm.weight_ih', att_w_hh='core.att_lstm.weight_hh', att_b_ih='core.att_lstm.bias_ih',
              att_b_hh='core.att_lstm.bias_hh', lang_w_ih='core.lang_lstm.weight_ih', lang_w_hh='core.lang_lstm.weight_hh',
              lang_b_ih='core.lang_lstm.bias_ih', lang_b_hh='core.lang_lstm.bias_hh',
              a1_w='core.attention.h2att.weight', a1_b='core.attention.h2att.bias',
              a1_aw='core.attention.alpha_net.weight', a1_ab='core.attention.alpha_net.bias',
              a2_w='core.attention2.h2att.weight', a2_b='core.attention2.h2att.bias',
              a2_aw='core.attention2.alpha_net.weight', a2_ab='core.attention2.alpha_net.bias')


@pytest.mark.parametrize('per_step', [True, False])
def test_bptt_matches_autograd_through_oracle(fake_kernels, per_step):
    W, pre, xt_all, att_mask, pnt_masks, Gh, Ga = _problem(3, per_step=per_step)
    B, Lc = xt_all.shape[:2]
    H = pre['fc'].shape[1]
    # --- autograd through the oracle's step
    Wr = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    prer = {k: v.clone().requires_grad_(True) for k, v in pre.items()}
    xr = xt_all.clone().requires_grad_(True)
    state = (torch.zeros(2, B, H, dtype=torch.float64), torch.zeros(2, B, H, dtype=torch.float64))
    outs, atts = [], []
    for t in range(Lc):
        pm = pnt_masks[:, t] if per_step else pnt_masks
        out, state, a2, _ = O.core_step(Wr, xr[:, t], prer, att_mask, pm, state)
        outs.append(out); atts.append(a2)
    h_ref, a_ref = torch.stack(outs, 1), torch.stack(atts, 1)
    ((h_ref * Gh).sum() + (a_ref * Ga).sum()).backward()
    # --- the hand-written Function
    keys = list(KEYMAP)
    Pm = [W[KEYMAP[k]].clone().requires_grad_(True) for k in keys]
    prem = {k: v.clone().requires_grad_(True) for k, v in pre.items()}
    xm = xt_all.clone().requires_grad_(True)
    h, a = decoder_bwd.DecoderLoopFn.apply(att_mask, pnt_masks, keys, prem['fc'], prem['conv'], prem['p_conv'],
                                           prem['pool'], prem['p_pool'], xm, *Pm)
    assert torch.allclose(h, h_ref, atol=1e-10) and torch.allclose(a, a_ref, atol=1e-6)
    ((h * Gh).sum() + (a * Ga).sum()).backward()
    for k in pre:
        assert torch.allclose(prem[k].grad, prer[k].grad, rtol=1e-8, atol=1e-10), k
    assert torch.allclose(xm.grad, xr.grad, rtol=1e-8, atol=1e-10)
    for k, p in zip(keys, Pm):
        assert torch.allclose(p.grad, Wr[KEYMAP[k]].grad, rtol=1e-8, atol=1e-10), k


@pytest.mark.parametrize('mode', ['featmap', 'region'])
def test_bptt_att_input_modes_match_autograd_through_oracle(fake_kernels, mode):
    """att_input_mode (opts.py:58, AttModel.py:140-151): 'featmap' feeds the frame-wise context alone to the language LSTM
    (the region attention still produces the grounding logits, so the region side gets a gradient through its scores only),
    'region' has no frame-wise side at all (its parameters: no gradient - None here as in autograd, so that an optimiser with
    weight decay skips them like the reference's does)."""
    W, pre, xt_all, att_mask, pnt_masks, Gh, Ga = _problem(5, per_step=True)
    B, Lc = xt_all.shape[:2]
    H = pre['fc'].shape[1]
    Wr = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    prer = {k: v.clone().requires_grad_(True) for k, v in pre.items()}
    xr = xt_all.clone().requires_grad_(True)
    state = (torch.zeros(2, B, H, dtype=torch.float64), torch.zeros(2, B, H, dtype=torch.float64))
    outs, atts = [], []
    for t in range(Lc):
        out, state, a2, _ = O.core_step(Wr, xr[:, t], dict(prer, att_input_mode=mode), att_mask, pnt_masks[:, t], state)
        outs.append(out); atts.append(a2)
    h_ref, a_ref = torch.stack(outs, 1), torch.stack(atts, 1)
    ((h_ref * Gh).sum() + (a_ref * Ga).sum()).backward()
    keys = list(KEYMAP)
    Pm = [W[KEYMAP[k]].clone().requires_grad_(True) for k in keys]
    prem = {k: v.clone().requires_grad_(True) for k, v in pre.items()}
    xm = xt_all.clone().requires_grad_(True)
    conv, p_conv = (prem['conv'], prem['p_conv']) if mode != 'region' else (torch.zeros(1, dtype=torch.float64),) * 2
    h, a = decoder_bwd.DecoderLoopFn.apply(att_mask, pnt_masks, (keys, mode), prem['fc'], conv, p_conv,
                                           prem['pool'], prem['p_pool'], xm, *Pm)
    assert torch.allclose(h, h_ref, atol=1e-10) and torch.allclose(a, a_ref, atol=1e-6)
    ((h * Gh).sum() + (a * Ga).sum()).backward()
    zero = lambda g: g is None or float(g.abs().max()) == 0.0
    for k in pre:
        if mode == 'region' and k in ('conv', 'p_conv'):
            assert prem[k].grad is None and prer[k].grad is None
            continue
        if prer[k].grad is None:                           # the region features under 'featmap': context unused
            assert mode == 'featmap' and k == 'pool' and zero(prem[k].grad)
            continue
        assert torch.allclose(prem[k].grad, prer[k].grad, rtol=1e-8, atol=1e-10), k
    assert torch.allclose(xm.grad, xr.grad, rtol=1e-8, atol=1e-10)
    for k, p in zip(keys, Pm):
        want = Wr[KEYMAP[k]].grad
        if want is None:                                    # the frame-wise attention's parameters under 'region'
            assert mode == 'region' and k.startswith('a1_') and p.grad is None, k        # None like the reference, not zeros
        else:
            assert torch.allclose(p.grad, want, rtol=1e-8, atol=1e-10), k


@pytest.mark.parametrize('region_mode', ['mix_mul', 'dp'])
def test_bptt_region_attn_modes_match_autograd_through_oracle(fake_kernels, region_mode):
    """region_attn_mode (opts.py:63, AttModel.py:82-95): the multiplicative score w . tanh(p * q) and the plain dot product
    p . q (no alpha_net in the module) through the hand-written BPTT."""
    W, pre, xt_all, att_mask, pnt_masks, Gh, Ga = _problem(6, per_step=True)
    if region_mode == 'dp':
        del W['core.attention2.alpha_net.weight'], W['core.attention2.alpha_net.bias']
    B, Lc = xt_all.shape[:2]
    H = pre['fc'].shape[1]
    Wr = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    prer = {k: v.clone().requires_grad_(True) for k, v in pre.items()}
    xr = xt_all.clone().requires_grad_(True)
    state = (torch.zeros(2, B, H, dtype=torch.float64), torch.zeros(2, B, H, dtype=torch.float64))
    outs, atts = [], []
    for t in range(Lc):
        out, state, a2, _ = O.core_step(Wr, xr[:, t], dict(prer, region_attn_mode=region_mode), att_mask, pnt_masks[:, t], state)
        outs.append(out); atts.append(a2)
    h_ref, a_ref = torch.stack(outs, 1), torch.stack(atts, 1)
    ((h_ref * Gh).sum() + (a_ref * Ga).sum()).backward()
    keys = [k for k in KEYMAP if KEYMAP[k] in W]
    Pm = [W[KEYMAP[k]].clone().requires_grad_(True) for k in keys]
    prem = {k: v.clone().requires_grad_(True) for k, v in pre.items()}
    xm = xt_all.clone().requires_grad_(True)
    h, a = decoder_bwd.DecoderLoopFn.apply(att_mask, pnt_masks, (keys, ('both', region_mode)), prem['fc'], prem['conv'],
                                           prem['p_conv'], prem['pool'], prem['p_pool'], xm, *Pm)
    assert torch.allclose(h, h_ref, atol=1e-10) and torch.allclose(a, a_ref, atol=1e-6)
    ((h * Gh).sum() + (a * Ga).sum()).backward()
    for k in pre:
        assert torch.allclose(prem[k].grad, prer[k].grad, rtol=1e-8, atol=1e-10), k
    assert torch.allclose(xm.grad, xr.grad, rtol=1e-8, atol=1e-10)
    for k, p in zip(keys, Pm):
        assert torch.allclose(p.grad, Wr[KEYMAP[k]].grad, rtol=1e-8, atol=1e-10), k
