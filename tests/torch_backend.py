"""TEST-ONLY torch stand-ins for the HIP kernel wrappers in gvd_amd.ops, with identical call signatures.
They let the CPU suite check the hand-written BPTT algebra (decoder_bwd.DecoderLoopFn) against autograd
through the oracle without a GPU.  Never imported by the product path."""
import torch

MIN_VALUE = -1e8


def gemm_nt(A, W, bias=None, act=0, out=None):
    y = A @ W.t()
    if bias is not None:
        y = y + bias
    if act:
        y = y.clamp(min=0)
    if out is not None:
        out.copy_(y)
        return out
    return y


def lstm_cell(xs, ws, h_prev, w_hh, b_ih, b_hh, c_prev, rowbias=None, gates_out=None, h_out=None, c_out=None):
    g = h_prev @ w_hh.t()
    for x, w in zip(xs, ws):
        g = g + x @ w.t()
    for b in (b_ih, b_hh, rowbias):
        if b is not None:
            g = g + b
    i, f, gg, o = g.chunk(4, 1)
    i, f, gg, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)
    c = f * c_prev + i * gg
    h = o * torch.tanh(c)
    if gates_out is not None:
        gates_out.copy_(torch.cat([i, f, gg, o], 1))
    if h_out is not None:
        h = h_out.copy_(h)
    if c_out is not None:
        c = c_out.copy_(c)
    return h, c


def _one_side(feats, p_feats, q, w, alpha_bias, att_mask=None, pnt_mask=None, logits_out=None, scores_out=None,
              score_mode=0):
    if score_mode == 2:                                   # 'dp' (AttModel.py:92-95)
        e = torch.bmm(p_feats, q.unsqueeze(2)).squeeze(2)
    else:                                                 # 'mix' / 'mix_mul' (AttModel.py:82-91)
        e = torch.tanh(p_feats * q.unsqueeze(1) if score_mode == 1 else p_feats + q.unsqueeze(1)) @ w + alpha_bias
    if att_mask is not None:
        e = e.masked_fill(att_mask.bool(), MIN_VALUE)
    if scores_out is not None:
        scores_out.copy_(e)
    if logits_out is not None:
        lo = e if pnt_mask is None else e.masked_fill(pnt_mask.bool(), MIN_VALUE)
        logits_out.copy_(lo)
    a = torch.softmax(e, 1)
    return torch.bmm(a.unsqueeze(1), feats).squeeze(1)


def attention_step(region, temporal, want_separate=False, out=None, cr_out=None, ct_out=None, sum_region=True):
    cr = _one_side(**region)
    ct = _one_side(**temporal) if temporal is not None else torch.zeros_like(cr)
    s = cr + ct if sum_region else ct.clone()
    if out is not None:
        s = out.copy_(s)
    if cr_out is not None:
        cr = cr_out.copy_(cr)
    if ct_out is not None:
        ct = ct_out.copy_(ct)
    return (s, cr, ct) if want_separate else s


def lstm_cell_bwd(dh, dc_next, gates, c_prev, c_new, dg_out=None, dh2=None):
    if dh2 is not None:
        dh = dh + dh2
    i, f, g, o = gates.chunk(4, 1)
    tc = torch.tanh(c_new)
    dc = dh * o * (1 - tc * tc)
    if dc_next is not None:
        dc = dc + dc_next
    dg = torch.cat([dc * g * i * (1 - i), dc * c_prev * f * (1 - f), dc * i * (1 - g * g), dh * tc * o * (1 - o)], 1)
    if dg_out is not None:
        dg = dg_out.copy_(dg)
    return dg, dc * f


def attn_bwd_chunks(N, B):
    return 3        # the HIP kernel's per-chunk partial slabs; any count checks the caller's reduction


def sum_chunks_pair(a, r, out):
    out[:, :a.shape[2]] = a.sum(1)
    out[:, a.shape[2]:] = r.sum(1)
    return out


def attn_bwd_step(side, alpha, ctx, d_ctx, d_logits=None, de_out=None, dq_out=None, dw_part=None, dab_part=None,
                  dq_part=None):
    feats, p_feats, q, w = side['feats'], side['p_feats'], side['q'], side['w']
    am, pm = side.get('att_mask'), side.get('pnt_mask')
    da = torch.bmm(feats, d_ctx.unsqueeze(2)).squeeze(2)
    dot = (ctx * d_ctx).sum(1, keepdim=True)
    de = alpha * (da - dot)
    if am is not None:
        de = de.masked_fill(am.bool(), 0.0)
    if d_logits is not None:
        keep = torch.ones_like(de, dtype=torch.bool)
        if am is not None:
            keep &= ~am.bool()
        if pm is not None:
            keep &= ~pm.bool()
        de = de + d_logits * keep
    mode = side.get('score_mode', 0)
    if mode == 2:
        dq = (de.unsqueeze(2) * p_feats).sum(1)
        dw, dab = torch.zeros_like(dq), torch.zeros_like(de.sum(1))
    else:
        t = torch.tanh(p_feats * q.unsqueeze(1) if mode == 1 else p_feats + q.unsqueeze(1))
        dq = ((de.unsqueeze(2) * w) * (1 - t * t) * (p_feats if mode == 1 else 1)).sum(1)
        dw = (de.unsqueeze(2) * t).sum(1)
        dab = de.sum(1)
    if de_out is not None:
        de = de_out.copy_(de)
    if dq_part is not None:         # left as per-chunk partials (uneven split) for sum_chunks_pair
        ncq = dq_part.shape[1]
        wq = torch.arange(1, ncq + 1, dtype=dq.dtype) / (ncq * (ncq + 1) / 2)
        dq = dq_part.copy_(dq.unsqueeze(1) * wq.view(1, ncq, 1))
    elif dq_out is not None:
        dq = dq_out.copy_(dq)
    if dw_part is not None:         # partial slabs: split the value unevenly over the chunks
        nc = dw_part.shape[1]
        wts = torch.arange(1, nc + 1, dtype=dw.dtype) / (nc * (nc + 1) / 2)
        dw = dw_part.copy_(dw.unsqueeze(1) * wts.view(1, nc, 1))
        dab = dab_part.copy_(dab.unsqueeze(1) * wts.view(1, nc))
    return de, dq, dw, dab


def attn_bwd_pfeats(p_feats, q_all, de_all, w, score_mode=0):
    out = torch.zeros_like(p_feats)
    for t in range(q_all.shape[0]):
        qt = q_all[t].unsqueeze(1)
        if score_mode == 2:
            out += de_all[t].unsqueeze(2) * qt
            continue
        th = torch.tanh(p_feats * qt if score_mode == 1 else p_feats + qt)
        out += de_all[t].unsqueeze(2) * w * (1 - th * th) * (qt if score_mode == 1 else 1)
    return out


def gru_layer(gi, w_f, b_f, w_b, b_b, B, T, Hh, flags=None, barrier=None):
    """Time loop of one bidirectional GRU layer; gi [B*T, 2*3*Hh] -> out [B,T,2*Hh] (gate order r,z,n)."""
    gi = gi.view(B, T, 2, 3 * Hh)
    out = torch.zeros(B, T, 2 * Hh, dtype=gi.dtype)
    for d, (w, b) in enumerate(((w_f, b_f), (w_b, b_b))):
        h = torch.zeros(B, Hh, dtype=gi.dtype)
        for t in (range(T) if d == 0 else range(T - 1, -1, -1)):
            gh = h @ w.t() + b
            g = gi[:, t, d]
            r = torch.sigmoid(g[:, :Hh] + gh[:, :Hh])
            z = torch.sigmoid(g[:, Hh:2 * Hh] + gh[:, Hh:2 * Hh])
            n = torch.tanh(g[:, 2 * Hh:] + r * gh[:, 2 * Hh:])
            h = (1 - z) * n + z * h
            out[:, t, d * Hh:(d + 1) * Hh] = h
    return out


def lstm_seq_layer(gi, w_f, b_f, w_b, b_b, B, T, Hh, flags=None, save=False):
    """Time loop of one bidirectional LSTM layer; gi [B*T, 2*4*Hh] -> out [B,T,2*Hh] (gate order i,f,g,o) and, with save,
    the post-activation gates [B,T,2,4*Hh] + cell states [B,T,2,Hh]."""
    gi = gi.view(B, T, 2, 4 * Hh)
    out = torch.zeros(B, T, 2 * Hh, dtype=gi.dtype)
    gates = torch.zeros(B, T, 2, 4 * Hh, dtype=gi.dtype)
    c_seq = torch.zeros(B, T, 2, Hh, dtype=gi.dtype)
    for d, (w, b) in enumerate(((w_f, b_f), (w_b, b_b))):
        h, c = torch.zeros(B, Hh, dtype=gi.dtype), torch.zeros(B, Hh, dtype=gi.dtype)
        for t in (range(T) if d == 0 else range(T - 1, -1, -1)):
            i, f, g, o = (gi[:, t, d] + h @ w.t() + b).chunk(4, 1)
            i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
            c = f * c + i * g
            h = o * torch.tanh(c)
            out[:, t, d * Hh:(d + 1) * Hh] = h
            gates[:, t, d] = torch.cat([i, f, g, o], 1)
            c_seq[:, t, d] = c
    return (out, gates, c_seq) if save else out


def gru_bwd_step(dout, gi, gh, out, carry_mm, carry_z, d_gi, d_gh, B, T, Hh, t_fw, t_bw, first):
    gi, gh = gi.view(B, T, 2, 3 * Hh), gh.view(B, T, 2, 3 * Hh)
    for d, t in ((0, t_fw), (1, t_bw)):
        tp = t + 1 if d else t - 1
        g, h = gi[:, t, d], gh[:, t, d]
        r = torch.sigmoid(g[:, :Hh] + h[:, :Hh])
        z = torch.sigmoid(g[:, Hh:2 * Hh] + h[:, Hh:2 * Hh])
        n = torch.tanh(g[:, 2 * Hh:] + r * h[:, 2 * Hh:])
        hp = out[:, tp, d * Hh:(d + 1) * Hh] if 0 <= tp < T else torch.zeros(B, Hh, dtype=gi.dtype)
        dh = dout[:, t, d * Hh:(d + 1) * Hh]
        if not first:
            dh = dh + carry_mm[d] + carry_z[d]
        dn = dh * (1 - z) * (1 - n * n)
        dz = dh * (hp - n) * z * (1 - z)
        dr = dn * h[:, 2 * Hh:] * r * (1 - r)
        d_gi[:, t, d] = torch.cat([dr, dz, dn], 1)
        d_gh[:, t, d] = torch.cat([dr, dz, dn * r], 1)
        carry_z[d] = dh * z


def dx_ok(M, Kred, ncols):
    return True


def dx_products(groups, M):
    """One 'launch' of up to 4 products out = A @ W (+ addend); every group reads its inputs before any output is written
    (the HIP kernel's groups run concurrently: an output of one group must not be an input of another)."""
    assert 1 <= len(groups) <= 4
    res = []
    for g in groups:
        assert g['A'].shape == (M, g['W'].shape[0])
        r = g['A'] @ g['W']
        if g.get('addend') is not None:
            r = r + g['addend']
        res.append(r)
    for g, r in zip(groups, res):
        g['out'].copy_(r)


def softmax_rows(x, out=None):
    return torch.softmax(x, dim=-1)


def rank_update(S, X, mask=None):
    if mask is not None:
        m = mask.bool()
        S = S.masked_fill(m if m.dim() == 3 else m.unsqueeze(1), 0.0)
    return torch.bmm(S.transpose(1, 2), X)
