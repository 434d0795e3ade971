"""Teacher-forced decoder loop (AttModel._forward token loop, model.py:421-453) on the HIP kernels.

`decoder_loop` runs Lc steps of TopDownCore.forward (AttModel.py:134-164): fused LSTM cells, one GEMM for
both attention queries, one streaming pass for both attentions.  Under autograd the loop is ONE
autograd.Function whose backward is a hand-scheduled BPTT (see `_DecoderLoopFn`), so no per-step
autograd graph, no saved [B,R,512] tanh tensors (SURVEY.md §7 "BPTT memory") and no per-step
[B,R,*] gradient buffers exist.
"""
import torch

from . import ops

K = ops   # kernel backend (the HIP library); tests substitute a torch stand-in to check the BPTT algebra on CPU


def _params(model):
    """The loop's parameters at the widths the kernels are built for (att_model.TopDownModel.core_params: `att_hid_size` /
    `input_encoding_size` below 512 enter zero-padded - exact - and autograd slices the gradients back)."""
    c = model.core
    P = dict(att_w_hh=c.att_lstm.weight_hh, att_b_ih=c.att_lstm.bias_ih,
             att_b_hh=c.att_lstm.bias_hh, lang_w_ih=c.lang_lstm.weight_ih, lang_w_hh=c.lang_lstm.weight_hh,
             lang_b_ih=c.lang_lstm.bias_ih, lang_b_hh=c.lang_lstm.bias_hh,
             a1_ab=c.attention.alpha_net.bias)
    if hasattr(c.attention2, 'alpha_net'):            # (none under region_attn_mode='dp', AttModel.py:63-66)
        P['a2_ab'] = c.attention2.alpha_net.bias
    P.update(model.core_params())
    return P


def forward_loop(P, fc, conv, p_conv, pool, p_pool, xt_all, att_mask, pnt_masks, save=None, mode='both'):
    """The forward recurrence.  `save` (dict) receives what the BPTT needs when given.  mode = att_input_mode
    (AttModel.py:140-151): 'region' runs no frame-wise side (conv / p_conv are not read), 'featmap' feeds the frame-wise
    context alone to the language LSTM (the region side still runs for the grounding logits)."""
    mode, region_mode = mode if isinstance(mode, tuple) else (mode, 'mix')      # (att_input_mode, region_attn_mode)
    use_t, sum_r = mode != 'region', mode != 'featmap'
    sm = {'mix': 0, 'mix_mul': 1, 'dp': 2}[region_mode]          # score function of the region side (AttModel.py:82-95)
    B, Lc, E = xt_all.shape
    H = fc.shape[1]
    A = p_pool.shape[2]
    R = pool.shape[1]
    dev = fc.device
    w_ih_fc, w_ih_xt = P['att_w_ih'][:, :H], P['att_w_ih'][:, H:]
    w_ih_att, w_ih_h = P['lang_w_ih'][:, :H], P['lang_w_ih'][:, H:]
    w_stack = torch.cat([P['a1_w'], P['a2_w']], 0)
    b_stack = torch.cat([P['a1_b'], P['a2_b']], 0)
    a1_aw, a2_aw = P['a1_aw'].reshape(-1), (P['a2_aw'].reshape(-1) if sm != 2 else None)
    # loop-invariant part of the att-LSTM gates: fc W_ih[:, :H]^T + b_ih + b_hh
    fc_gates = K.gemm_nt(fc, w_ih_fc, P['att_b_ih']) + P['att_b_hh']
    h_att = torch.zeros(B, H, device=dev, dtype=fc.dtype); c_att = torch.zeros(B, H, device=dev, dtype=fc.dtype)
    h_lang = torch.zeros(B, H, device=dev, dtype=fc.dtype); c_lang = torch.zeros(B, H, device=dev, dtype=fc.dtype)
    h_all = torch.empty(B, Lc, H, device=dev, dtype=fc.dtype)
    att2_w = torch.empty(B, Lc, R, device=dev, dtype=fc.dtype)
    am = att_mask[:, 1:]
    per_step_mask = pnt_masks.dim() == 3
    if save is not None:
        save.update(gates_att=torch.empty(Lc, B, 4 * H, device=dev, dtype=fc.dtype), gates_lang=torch.empty(Lc, B, 4 * H, device=dev, dtype=fc.dtype),
                    c_att=torch.empty(Lc + 1, B, H, device=dev, dtype=fc.dtype), c_lang=torch.empty(Lc + 1, B, H, device=dev, dtype=fc.dtype),
                    h_att=torch.empty(Lc + 1, B, H, device=dev, dtype=fc.dtype), h_lang=torch.empty(Lc + 1, B, H, device=dev, dtype=fc.dtype),
                    q12=torch.empty(Lc, B, 2 * A, device=dev, dtype=fc.dtype), att_sum=torch.empty(Lc, B, H, device=dev, dtype=fc.dtype),
                    ctx_r=torch.empty(Lc, B, H, device=dev, dtype=fc.dtype), ctx_t=torch.empty(Lc, B, H, device=dev, dtype=fc.dtype),
                    scores_r=torch.empty(B, Lc, R, device=dev, dtype=fc.dtype),
                    scores_t=torch.empty(B, Lc, conv.shape[1] if use_t else 0, device=dev, dtype=fc.dtype),
                    w_stack=w_stack)
        for k in ('c_att', 'c_lang', 'h_att', 'h_lang'):
            save[k][0].zero_()
    if save is not None:
        # training: every kernel writes its step's slice of the saved-state arrays in place (no per-step copies)
        for t in range(Lc):
            h_att, c_att = K.lstm_cell([xt_all[:, t]], [w_ih_xt], save['h_att'][t], P['att_w_hh'], None, None,
                                       save['c_att'][t], rowbias=fc_gates, gates_out=save['gates_att'][t],
                                       h_out=save['h_att'][t + 1], c_out=save['c_att'][t + 1])
            q12 = K.gemm_nt(h_att, w_stack, b_stack, out=save['q12'][t])
            pmask = (pnt_masks[:, t] if per_step_mask else pnt_masks)[:, 1:]
            region = dict(feats=pool, p_feats=p_pool, q=q12[:, A:], w=a2_aw, alpha_bias=P.get('a2_ab'), att_mask=am,
                          pnt_mask=pmask, logits_out=att2_w[:, t], scores_out=save['scores_r'][:, t], score_mode=sm)
            temporal = dict(feats=conv, p_feats=p_conv, q=q12[:, :A], w=a1_aw, alpha_bias=P['a1_ab'],
                            scores_out=save['scores_t'][:, t]) if use_t else None
            att_sum, _, _ = K.attention_step(region, temporal, want_separate=True, out=save['att_sum'][t],
                                             cr_out=save['ctx_r'][t], ct_out=save['ctx_t'][t] if (use_t and sum_r) else None,
                                             sum_region=sum_r)          # ('featmap': att_sum[t] IS the temporal context)
            K.lstm_cell([att_sum, h_att], [w_ih_att, w_ih_h], save['h_lang'][t], P['lang_w_hh'], P['lang_b_ih'],
                        P['lang_b_hh'], save['c_lang'][t], gates_out=save['gates_lang'][t],
                        h_out=save['h_lang'][t + 1], c_out=save['c_lang'][t + 1])
        h_all.copy_(save['h_lang'][1:].transpose(0, 1))
        return h_all, att2_w
    for t in range(Lc):
        xt = xt_all[:, t]
        h_att, c_att = K.lstm_cell([xt], [w_ih_xt], h_att, P['att_w_hh'], None, None, c_att, rowbias=fc_gates)
        q12 = K.gemm_nt(h_att, w_stack, b_stack)
        pmask = (pnt_masks[:, t] if per_step_mask else pnt_masks)[:, 1:]
        region = dict(feats=pool, p_feats=p_pool, q=q12[:, A:], w=a2_aw, alpha_bias=P.get('a2_ab'), att_mask=am,
                      pnt_mask=pmask, logits_out=att2_w[:, t], score_mode=sm)
        temporal = dict(feats=conv, p_feats=p_conv, q=q12[:, :A], w=a1_aw, alpha_bias=P['a1_ab']) if use_t else None
        att_sum = K.attention_step(region, temporal, sum_region=sum_r)
        h_lang, c_lang = K.lstm_cell([att_sum, h_att], [w_ih_att, w_ih_h], h_lang, P['lang_w_hh'],
                                     P['lang_b_ih'], P['lang_b_hh'], c_lang, h_out=h_all[:, t])
    return h_all, att2_w


def decoder_loop(model, pre, xt_all, att_mask, pnt_masks):
    P = _params(model)
    mode = (getattr(model, 'att_input_mode', 'both'), getattr(model, 'region_attn_mode', 'mix'))
    conv, p_conv = pre['conv'], pre['p_conv']
    if mode[0] == 'region':
        # no frame-wise side: one-element stand-ins keep the autograd.Function's argument list (nothing reads them)
        conv = p_conv = pre['fc'].new_zeros(1)
    tensors = [pre['fc'], conv, p_conv, pre['pool'], pre['p_pool'], xt_all] + list(P.values())
    if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
        from .decoder_bwd import DecoderLoopFn
        keys = list(P.keys())
        return DecoderLoopFn.apply(att_mask, pnt_masks, (keys, mode), pre['fc'], conv, p_conv, pre['pool'],
                                   pre['p_pool'], xt_all.contiguous(), *[P[k] for k in keys])
    Pd = {k: v.detach() for k, v in P.items()}
    return forward_loop(Pd, pre['fc'].detach(), conv.detach(), p_conv.detach(), pre['pool'].detach(),
                        pre['p_pool'].detach(), xt_all.detach().contiguous(), att_mask, pnt_masks, mode=mode)
