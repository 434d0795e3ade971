"""Hand-scheduled BPTT of the teacher-forced decoder loop as ONE autograd.Function.

Forward = decoder_fn.forward_loop (HIP kernels), keeping only O(B*H) state per step (gates, cell states,
queries, attention scores/contexts).  Backward walks the steps in reverse:
  * pointwise LSTM backward kernel, dX GEMMs (library GEMMs) for the recurrent/input gradients;
  * per step and attention side ONE streaming pass over feats/p_feats (gvd_attn_bwd_step) that recomputes
    tanh, applies the softmax/mask backward and yields de[n], d_q, d_w, d_alpha_bias;
  * after the loop, the gradients w.r.t. the big per-segment tensors are formed ONCE for all steps:
    d_pool/d_conv = alpha^T d_ctx (batched GEMM), d_p_pool/d_p_conv by gvd_attn_bwd_pfeats, and every weight
    gradient as one GEMM over the stacked [Lc*B, .] activations.
Reference semantics: autograd through AttModel.py:134-164 x Lc (model.py:421-453).
"""
import torch
import torch.nn.functional as F

from . import decoder_fn


def _tn(dY, X):
    """dY^T X (a weight gradient over the stacked [Lc*B, .] activations) on the K-strided MFMA kernel; a contraction length
    that is not a multiple of its 32-deep k tile (odd batch sizes) is zero-padded first.  (The torch stand-in backend of the
    CPU tests has no gemm_dw: plain matmul there.)"""
    f = getattr(decoder_fn.K, 'gemm_dw', None)
    if f is None:
        return dY.t() @ X
    pad = (-dY.shape[0]) % 32
    if pad:
        dY, X = F.pad(dY, (0, 0, 0, pad)), F.pad(X, (0, 0, 0, pad))
    r = f(dY.contiguous(), X.contiguous())
    if r is None:
        decoder_fn.K.library_fallback('token-loop dW', '%s^T x %s' % (tuple(dY.shape), tuple(X.shape)))
        return dY.t() @ X
    return r


def _dx(K, groups, M):
    """The products out_g = A_g @ W_g (+ addend_g) of one BPTT step in one launch (ops.dx_products) when every group has a
    shape the kernel takes; one matmul per group otherwise (the stand-in backend, widths that are not multiples of 128)."""
    f = getattr(K, 'dx_products', None)
    if f is not None and all(K.dx_ok(M, g['W'].shape[0], g['W'].shape[1]) for g in groups):
        f(groups, M)
        return
    if f is not None:
        K.library_fallback('token-loop dX', ', '.join(str(tuple(g['W'].shape)) for g in groups))
    for g in groups:
        if g.get('addend') is not None:
            torch.addmm(g['addend'], g['A'], g['W'], out=g['out'])
        else:
            torch.mm(g['A'], g['W'], out=g['out'])


class DecoderLoopFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, att_mask, pnt_masks, keys, fc, conv, p_conv, pool, p_pool, xt_all, *params):
        keys, mode = keys if isinstance(keys, tuple) else (keys, 'both')        # (parameter names, att_input_mode)
        P = dict(zip(keys, params))
        save = {}
        h_all, att2_w = decoder_fn.forward_loop(P, fc, conv, p_conv, pool, p_pool, xt_all, att_mask, pnt_masks,
                                                save=save, mode=mode)
        ctx.keys = keys
        ctx.mode = mode
        ctx.save = save
        ctx.masks = (att_mask, pnt_masks)
        ctx.save_for_backward(fc, conv, p_conv, pool, p_pool, xt_all, *params)
        return h_all, att2_w

    @staticmethod
    def backward(ctx, d_h_all, d_att2w):
        K = decoder_fn.K
        fc, conv, p_conv, pool, p_pool, xt_all = ctx.saved_tensors[:6]
        P = dict(zip(ctx.keys, ctx.saved_tensors[6:]))
        S = ctx.save
        att_mask, pnt_masks = ctx.masks
        B, Lc, E = xt_all.shape
        H = fc.shape[1]
        A = p_pool.shape[2]
        # att_input_mode (AttModel.py:140-151).  'region': no frame-wise side - its scores, queries and parameters get no
        # gradient (None, like the reference's .grad = None).  'featmap': the region context is not an input of the
        # language LSTM - the region side's d_ctx is zero (its scores still carry the grounding losses' gradient) and the
        # region features get a gradient through the projection only
        mode, region_mode = ctx.mode if isinstance(ctx.mode, tuple) else (ctx.mode, 'mix')
        use_t, sum_r = mode != 'region', mode != 'featmap'
        sm = {'mix': 0, 'mix_mul': 1, 'dp': 2}[region_mode]      # score function of the region side (AttModel.py:82-95)
        R, Ft = pool.shape[1], (conv.shape[1] if use_t else 0)
        dev = fc.device
        per_step_mask = pnt_masks.dim() == 3
        am = att_mask[:, 1:]
        w_stack = S['w_stack']
        a1_aw, a2_aw = P['a1_aw'].reshape(-1), (P['a2_aw'].reshape(-1) if sm != 2 else None)
        softmax = getattr(K, 'softmax_rows', None) or (lambda x: torch.softmax(x, dim=-1))
        alpha_r = softmax(S['scores_r'])                        # [B,Lc,R]
        alpha_t = softmax(S['scores_t']) if use_t else None     # [B,Lc,Ft]
        if d_h_all is None:
            d_h_all = torch.zeros(B, Lc, H, device=dev, dtype=fc.dtype)
        dG_lang = torch.empty(Lc, B, 4 * H, device=dev, dtype=fc.dtype)
        dG_att = torch.empty(Lc, B, 4 * H, device=dev, dtype=fc.dtype)
        dq12_all = torch.empty(Lc, B, 2 * A, device=dev, dtype=fc.dtype)
        dX_all = torch.empty(Lc, B, 2 * H, device=dev, dtype=fc.dtype)      # [d(att+att2) | d h_att] of the lang-LSTM input
        de_r_all = torch.empty(Lc, B, R, device=dev, dtype=fc.dtype)
        de_t_all = torch.empty(Lc, B, Ft, device=dev, dtype=fc.dtype)
        # per-chunk partials of the alpha_net gradients of every step: reduced ONCE after the loop
        nc_r, nc_t = K.attn_bwd_chunks(R, B), (K.attn_bwd_chunks(Ft, B) if use_t else 1)
        dw_r_all = torch.empty(Lc, B, nc_r, A, device=dev, dtype=fc.dtype)
        dw_t_all = torch.empty(Lc, B, nc_t, A, device=dev, dtype=fc.dtype)
        dab_r_all = torch.empty(Lc, B, nc_r, device=dev, dtype=fc.dtype)
        dab_t_all = torch.empty(Lc, B, nc_t, device=dev, dtype=fc.dtype)
        # per-chunk partials of one step's two query gradients (rewritten every step; summed into dq12_all[t] by ONE launch)
        dq_r_part = torch.empty(B, nc_r, A, device=dev, dtype=fc.dtype)
        dq_t_part = (torch.empty if use_t else torch.zeros)(B, nc_t, A, device=dev, dtype=fc.dtype)
        zero_ctx = None if sum_r else torch.zeros(B, H, device=dev, dtype=fc.dtype)
        dh_att_next = dh_lang_next = None
        dc_att_next = dc_lang_next = None
        w_lang_ih, w_lang_hh, w_att_hh = P['lang_w_ih'], P['lang_w_hh'], P['att_w_hh']
        # recurrent hidden-state gradients of the two cells: ping-pong buffers the step's product launch writes
        dh_lang_buf = [torch.empty(B, H, device=dev, dtype=fc.dtype) for _ in range(2)]
        dh_att_buf = [torch.empty(B, H, device=dev, dtype=fc.dtype) for _ in range(2)]
        dh_att_cur = torch.empty(B, H, device=dev, dtype=fc.dtype)
        dg_att_prev = None                                       # gate gradients of the attention cell at step t + 1
        for t in range(Lc - 1, -1, -1):
            # every kernel writes its step's slice of the [Lc, ...] arrays in place: no per-step copies / partial sums
            # (the two addends of a hidden-state gradient - this step's output gradient and the recurrent term of step
            # t + 1 - are added inside the pointwise kernel)
            dg, dc_lang_next = K.lstm_cell_bwd(d_h_all[:, t], dc_lang_next, S['gates_lang'][t], S['c_lang'][t],
                                               S['c_lang'][t + 1], dg_out=dG_lang[t], dh2=dh_lang_next)
            # ONE launch: [d(att+att2) | d h_att] = dg W_ih(lang), d h_lang(t-1) = dg W_hh(lang) and - from the attention
            # cell's gate gradients of step t + 1 - d h_att(t) = dg_att W_hh(att)
            dX = dX_all[t]
            dh_lang_next = dh_lang_buf[t & 1]
            groups = [dict(A=dg, W=w_lang_ih, out=dX), dict(A=dg, W=w_lang_hh, out=dh_lang_next)]
            if dg_att_prev is not None:
                dh_att_next = dh_att_buf[t & 1]
                groups.append(dict(A=dg_att_prev, W=w_att_hh, out=dh_att_next))
            _dx(K, groups, B)
            d_att_sum = dX[:, :H]
            pmask = (pnt_masks[:, t] if per_step_mask else pnt_masks)[:, 1:]
            q12 = S['q12'][t]
            region = dict(feats=pool, p_feats=p_pool, q=q12[:, A:], w=a2_aw, alpha_bias=P.get('a2_ab'), att_mask=am,
                          pnt_mask=pmask, score_mode=sm)
            temporal = dict(feats=conv, p_feats=p_conv, q=q12[:, :A], w=a1_aw, alpha_bias=P['a1_ab']) if use_t else None
            dl = d_att2w[:, t] if d_att2w is not None else None
            dq12 = dq12_all[t]
            K.attn_bwd_step(region, alpha_r[:, t], S['ctx_r'][t], d_att_sum if sum_r else zero_ctx, dl, de_out=de_r_all[t],
                            dq_part=dq_r_part, dw_part=dw_r_all[t], dab_part=dab_r_all[t])
            if use_t:
                K.attn_bwd_step(temporal, alpha_t[:, t], (S['ctx_t'] if sum_r else S['att_sum'])[t], d_att_sum, None, de_out=de_t_all[t],
                                dq_part=dq_t_part, dw_part=dw_t_all[t], dab_part=dab_t_all[t])
            K.sum_chunks_pair(dq_t_part, dq_r_part, dq12)        # [:, :A] temporal (a1), [:, A:] region (a2)
            _dx(K, [dict(A=dq12, W=w_stack, out=dh_att_cur, addend=dX[:, H:])], B)      # d h_att = dX[:, H:] + dq12 W_h2att
            dg_att_prev, dc_att_next = K.lstm_cell_bwd(dh_att_cur, dc_att_next, S['gates_att'][t], S['c_att'][t],
                                                       S['c_att'][t + 1], dg_out=dG_att[t], dh2=dh_att_next)
        dw_r, dab_r = dw_r_all.sum((0, 1, 2)), dab_r_all.sum().view(1)
        if use_t:
            dw_t, dab_t = dw_t_all.sum((0, 1, 2)), dab_t_all.sum().view(1)
        else:
            dw_t, dab_t = torch.zeros(A, device=dev, dtype=fc.dtype), torch.zeros(1, device=dev, dtype=fc.dtype)
        dctx_all = dX_all[:, :, :H]

        # ---- gradients formed once for all steps
        dGa = dG_att.view(Lc * B, 4 * H)
        dGl = dG_lang.view(Lc * B, 4 * H)
        xt_flat = xt_all.transpose(0, 1).reshape(Lc * B, E)
        h_att_prev = S['h_att'][:Lc].reshape(Lc * B, H)
        h_att_new = S['h_att'][1:].reshape(Lc * B, H)
        h_lang_prev = S['h_lang'][:Lc].reshape(Lc * B, H)
        att_sum = S['att_sum'].view(Lc * B, H)
        w_att_ih = P['att_w_ih']
        sumG = dG_att.sum(0)                                     # [B,4H]: the fc part of the input is loop invariant
        g = {}
        g['fc'] = torch.empty(B, H, device=dev, dtype=fc.dtype)
        _dx(K, [dict(A=sumG, W=w_att_ih[:, :H], out=g['fc'])], B)
        dxt = torch.empty(Lc * B, E, device=dev, dtype=fc.dtype)
        _dx(K, [dict(A=dGa, W=w_att_ih[:, H:], out=dxt)], Lc * B)
        g['xt_all'] = dxt.view(Lc, B, E).transpose(0, 1).contiguous()
        g['att_w_ih'] = torch.cat([_tn(sumG, fc), _tn(dGa, xt_flat)], dim=1)
        g['att_w_hh'] = _tn(dGa, h_att_prev)
        g['att_b_ih'] = dGa.sum(0)
        g['att_b_hh'] = g['att_b_ih']
        g['lang_w_ih'] = torch.cat([_tn(dGl, att_sum), _tn(dGl, h_att_new)], dim=1)
        g['lang_w_hh'] = _tn(dGl, h_lang_prev)
        g['lang_b_ih'] = dGl.sum(0)
        g['lang_b_hh'] = g['lang_b_ih']
        dq_flat = dq12_all.view(Lc * B, 2 * A)
        d_wstack = _tn(dq_flat, h_att_new)                       # [2A,H]
        d_bstack = dq_flat.sum(0)
        g['a1_w'], g['a2_w'] = d_wstack[:A], d_wstack[A:]
        g['a1_b'], g['a2_b'] = d_bstack[:A], d_bstack[A:]
        g['a1_aw'], g['a2_aw'] = dw_t.view(1, A), dw_r.view(1, A)      # (a2_*: not among ctx.keys under 'dp' - no alpha_net)
        g['a1_ab'], g['a2_ab'] = dab_t, dab_r
        dctx_b = dctx_all.transpose(0, 1)                        # [B,Lc,H] view of dX_all[:, :, :H]
        ru = getattr(K, 'rank_update_any', None)
        if ru is None or H % 128 != 0:
            if ru is not None:
                K.library_fallback('alpha^T d_ctx', 'H = %d' % H)
            ru = lambda alpha, d: torch.bmm(alpha.transpose(1, 2), d)            # noqa: E731
        # alpha^T d_ctx over all steps: one streaming write of [B,R,H] / [B,Ft,H] (csrc/stream_mm.hip)
        g['pool'] = ru(alpha_r, dctx_b) if sum_r else torch.zeros_like(pool)     # [B,R,H]
        g['conv'] = ru(alpha_t, dctx_b) if use_t else None
        g['p_pool'] = K.attn_bwd_pfeats(p_pool, S['q12'][:, :, A:], de_r_all, a2_aw, score_mode=sm)
        g['p_conv'] = K.attn_bwd_pfeats(p_conv, S['q12'][:, :, :A], de_t_all, a1_aw) if use_t else None
        if not use_t:
            # 'region': the frame-wise attention module is never called - the reference leaves its parameters' .grad = None (an
            # optimiser with weight decay skips them, its state holds no entry for them): None here too, not zeros
            for k in ('a1_w', 'a1_b', 'a1_aw', 'a1_ab'):
                g[k] = None
        ctx.save = None
        names = ['fc', 'conv', 'p_conv', 'pool', 'p_pool', 'xt_all'] + list(ctx.keys)
        out = [None, None, None]
        for i, n in enumerate(names):
            out.append(g[n] if ctx.needs_input_grad[3 + i] else None)
        return tuple(out)
