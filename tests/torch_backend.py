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


def lstm_cell(xs, ws, h_prev, w_hh, b_ih, b_hh, c_prev, rowbias=None, gates_out=None):
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
    return h, c


def _one_side(feats, p_feats, q, w, alpha_bias, att_mask=None, pnt_mask=None, logits_out=None, scores_out=None):
    e = torch.tanh(p_feats + q.unsqueeze(1)) @ w + alpha_bias
    if att_mask is not None:
        e = e.masked_fill(att_mask.bool(), MIN_VALUE)
    if scores_out is not None:
        scores_out.copy_(e)
    if logits_out is not None:
        lo = e if pnt_mask is None else e.masked_fill(pnt_mask.bool(), MIN_VALUE)
        logits_out.copy_(lo)
    a = torch.softmax(e, 1)
    return torch.bmm(a.unsqueeze(1), feats).squeeze(1)


def attention_step(region, temporal, want_separate=False):
    cr = _one_side(**region)
    ct = _one_side(**temporal) if temporal is not None else torch.zeros_like(cr)
    return (cr + ct, cr, ct) if want_separate else cr + ct


def lstm_cell_bwd(dh, dc_next, gates, c_prev, c_new):
    i, f, g, o = gates.chunk(4, 1)
    tc = torch.tanh(c_new)
    dc = dh * o * (1 - tc * tc)
    if dc_next is not None:
        dc = dc + dc_next
    dg = torch.cat([dc * g * i * (1 - i), dc * c_prev * f * (1 - f), dc * i * (1 - g * g), dh * tc * o * (1 - o)], 1)
    return dg, dc * f


def attn_bwd_step(side, alpha, ctx, d_ctx, d_logits=None):
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
    t = torch.tanh(p_feats + q.unsqueeze(1))
    dq = ((de.unsqueeze(2) * w) * (1 - t * t)).sum(1)
    dw = (de.unsqueeze(2) * t).sum(1)
    return de, dq, dw, de.sum(1)


def attn_bwd_pfeats(p_feats, q_all, de_all, w):
    out = torch.zeros_like(p_feats)
    for t in range(q_all.shape[0]):
        th = torch.tanh(p_feats + q_all[t].unsqueeze(1))
        out += de_all[t].unsqueeze(2) * w * (1 - th * th)
    return out
