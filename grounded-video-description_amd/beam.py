"""Beam-search decode on the HIP path, batched over samples x beams.

Semantics = AttModel._sample_beam + CaptionModel.beam_search (model.py:627-742; CaptionModelBU.py:24-185)
with the minimal repair documented in SURVEY.md §3.4 (the reference's own beam path raises a TypeError as shipped;
parity is pinned "reference-with-shim": tests/golden/beam*.npz hold the outputs of the reference's own beam_search
run under a run-time shim that drops the two stray core arguments, oracle/ref_harness.beam_shim).
Reproduced: candidate order word-rank-major / beam-minor with a stable sort by descending summed log-prob;
only beam 0 expands at t=0; no UNK suppression; finished beams get sum = -1000 but stay in the pool; the core
also runs after the last token; att2 holds argmax-over-all-regions indices and a finished beam's att2 is the
FINAL content of its beam column (the reference stores a view, CaptionModelBU.py:159).

MI355X design: the reference loops over samples one by one and expands every per-sample tensor to
`beam_size` copies (model.py:708-726), with a device->host copy and a Python sort per step
(CaptionModelBU.py:129,61).  Here all B*K beam rows advance together: one LSTM/logit GEMM per step over
B*K rows, the attention kernel reads each sample's region features ONCE for its K beams (`group=K`
indirection, no expanded copies), top-K per row is a HIP kernel, and the K*K candidate merge + state
gathers are a handful of batched device ops — no host synchronisation inside the loop.
"""
import ctypes as C
import os

import torch

from . import ops
from .hip import BeamStepArgs, check, lib, ptr, stream_ptr


_STATE = ('h_att', 'c_att', 'h_lang', 'c_lang')


def _state(stack):
    """The four recurrent states as views of ONE [4, rows, H] tensor: forking the beams' states from their parents is then
    one gather launch per step instead of four."""
    return dict(zip(_STATE, stack.unbind(0)), stack=stack)


def _fork(st, parent):
    return _state(st['stack'].index_select(1, parent))


def _core_rows(P, st, xt, fc_gates, pre, pmask_rows, K, att2_out):
    """One TopDownCore step (AttModel.py:134-164) for B*K beam rows; st = _state(...) of (h_att, c_att, h_lang, c_lang)."""
    H = st['h_att'].shape[1]
    A = pre['p_pool'].shape[2]
    nxt = torch.empty_like(st['stack'])
    h_att, c_att = ops.lstm_cell([xt], [P['att_w_ih'][:, H:]], st['h_att'], P['att_w_hh'], None, None, st['c_att'],
                                 rowbias=fc_gates, h_out=nxt[0], c_out=nxt[1])
    q12 = ops.gemm_nt(h_att, P['w_stack'], P['b_stack'])
    sm = ops.SCORE_MODES[pre.get('region_attn_mode', 'mix')]           # AttModel.py:82-95; 'dp' has no alpha_net
    region = dict(feats=pre['pool'], p_feats=pre['p_pool'], q=q12[:, A:], w=P['att2_alpha_w'].view(-1) if sm != 2 else None,
                  alpha_bias=P.get('att2_alpha_b'), att_mask=pmask_rows[:, 1:], pnt_mask=pmask_rows[:, 1:],
                  logits_out=att2_out, group=K, score_mode=sm)
    # att_input_mode (AttModel.py:140-151): 'region' has no frame-wise side, 'featmap' feeds the frame-wise context alone
    temporal = None if pre['conv'] is None else dict(feats=pre['conv'], p_feats=pre['p_conv'], q=q12[:, :A],
                                                     w=P['att1_alpha_w'].view(-1), alpha_bias=P['att1_alpha_b'], group=K)
    att_sum = ops.attention_step(region, temporal, sum_region=pre.get('att_input_mode', 'both') != 'featmap')
    ops.lstm_cell([att_sum, h_att], [P['lang_w_ih'][:, :H], P['lang_w_ih'][:, H:]], st['h_lang'],
                  P['lang_w_hh'], P['lang_b_ih'], P['lang_b_hh'], st['c_lang'], h_out=nxt[2], c_out=nxt[3])
    return _state(nxt)


def beam_decode(model, pre, P, K, fused_step=True):
    """-> seq i64 [B,L], seqLogprobs f32 [B,L], att2 i64 [B,L] (global region argmax per step)."""
    fc = pre['fc']
    B, H = fc.shape
    R = pre['pool'].shape[1]
    L = model.seq_length
    dev = fc.device
    rows = B * K
    P = dict(P)
    P['w_stack'] = torch.cat([P['att1_h2att_w'], P['att2_h2att_w']], 0)
    P['b_stack'] = torch.cat([P['att1_h2att_b'], P['att2_h2att_b']], 0)
    fc_gates = (ops.gemm_nt(fc, P['att_w_ih'][:, :H], P['att_b_ih']) + P['att_b_hh']).repeat_interleave(K, 0)
    pm_rows = pre['pnt_mask'].repeat_interleave(K, 0).contiguous()
    att2_w = torch.empty(rows, R, device=dev)
    st = _state(torch.zeros(4, rows, H, device=dev))
    it = torch.zeros(rows, dtype=torch.int64, device=dev)
    st = _core_rows(P, st, ops.embed_relu(it, P['embed']), fc_gates, pre, pm_rows, K, att2_w)   # BOS step
    att2_first = att2_w.view(B, K, R)[:, 0].max(dim=1)[1]                                         # model.py:733
    att2_ind = torch.full((B, K), -1, dtype=torch.int64, device=dev)

    beam_seq = torch.zeros(L, B, K, dtype=torch.int64, device=dev)
    beam_lps = torch.zeros(L, B, K, device=dev)
    beam_att = torch.full((L, B, K), -1, dtype=torch.int64, device=dev)
    sums = torch.zeros(B, K, device=dev)
    best_p = torch.full((B,), float('-inf'), device=dev)
    best_seq = torch.zeros(B, L, dtype=torch.int64, device=dev)
    best_lps = torch.zeros(B, L, device=dev)
    best_vix = torch.zeros(B, dtype=torch.int64, device=dev)
    base = (torch.arange(B, device=dev) * K).view(B, 1)
    kk = torch.arange(K, device=dev)

    # beam widths the one-wave-per-sample step kernel holds (csrc/beam.hip: K <= 8) take it; wider beams - and tests that
    # pass fused_step=False - the batched torch formulation of the same bookkeeping below
    fused = fused_step and K <= 8
    if fused:
        parent = torch.empty(rows, dtype=torch.int64, device=dev)
        word_rows = torch.empty(rows, dtype=torch.int64, device=dev)
        a = BeamStepArgs()
        a.sums, a.beam_seq, a.beam_lps, a.beam_att = ptr(sums), ptr(beam_seq), ptr(beam_lps), ptr(beam_att)
        a.best_p, a.best_seq, a.best_lps, a.best_vix = ptr(best_p), ptr(best_seq), ptr(best_lps), ptr(best_vix)
        a.parent, a.word = ptr(parent), ptr(word_rows)
        a.B, a.K, a.L = B, K, L
    for t in range(L if fused else 0):
        # one launch for the candidate merge / history fork / finished-beam record of all samples (csrc/beam.hip)
        logits = ops.gemm_nt(st['h_lang'], P['logit_w'], P['logit_b'])
        _, _, ys, ix = ops.logsoftmax_rows(logits, topk=K)
        a.ys, a.ix, a.att2_ind, a.t = ptr(ys), ptr(ix), ptr(att2_ind), t
        check(lib().gvd_beam_step(C.byref(a), stream_ptr()), 'gvd_beam_step')
        st = _fork(st, parent)
        st = _core_rows(P, st, ops.embed_relu(word_rows, P['embed']), fc_gates, pre, pm_rows, K, att2_w)
        att2_ind = att2_w.view(B, K, R).max(dim=2)[1].contiguous()   # CaptionModelBU.py:182

    for t in range(0 if fused else L):
        logits = ops.gemm_nt(st['h_lang'], P['logit_w'], P['logit_b'])
        _, _, ys, ix = ops.logsoftmax_rows(logits, topk=K)          # sorted log-probs / word ids per beam row
        ys, ix = ys.view(B, K, K), ix.view(B, K, K)                  # [b, q, c]
        # candidates in the reference's order j = c*rows_t + q  (CaptionModelBU.py:49-55)
        if t == 0:
            cand_p = sums[:, 0:1] + ys[:, 0, :]                      # only beam 0 expands: [B, K(c)]
            order = torch.sort(cand_p, dim=1, descending=True, stable=True)[1][:, :K]
            q_sel = torch.zeros(B, K, dtype=torch.int64, device=dev)
            c_sel = order
        else:
            cand_p = (sums.unsqueeze(2) + ys).transpose(1, 2).reshape(B, K * K)   # index c*K + q, fp32 sums
            order = torch.sort(cand_p, dim=1, descending=True, stable=True)[1][:, :K]
            q_sel, c_sel = order % K, order // K
        new_p = torch.gather(cand_p, 1, order)
        word = ix[torch.arange(B, device=dev).view(B, 1), q_sel, c_sel]
        r = ys[torch.arange(B, device=dev).view(B, 1), q_sel, c_sel]
        # fork histories / state from parent beam q (CaptionModelBU.py:75-96)
        if t >= 1:
            gi = q_sel.unsqueeze(0).expand(t, B, K)
            beam_seq[:t] = torch.gather(beam_seq[:t], 2, gi)
            beam_lps[:t] = torch.gather(beam_lps[:t], 2, gi)
            beam_att[:t] = torch.gather(beam_att[:t], 2, gi)
            beam_att[t] = torch.gather(att2_ind, 1, q_sel)
        parent = (base + q_sel).view(-1)
        st = _fork(st, parent)
        beam_seq[t] = word
        beam_lps[t] = r
        sums = new_p
        # finished beams (CaptionModelBU.py:154-166): recorded in vix order, best = first maximum of p
        fin = (word == 0) if t < L - 1 else torch.ones_like(word, dtype=torch.bool)
        p_fin = torch.where(fin, sums, torch.full_like(sums, float('-inf')))
        step_best, _ = p_fin.max(dim=1)
        first = torch.where(p_fin == step_best.unsqueeze(1), kk.view(1, K), torch.full_like(word, K)).min(dim=1)[0]
        first = first.clamp(max=K - 1)
        better = step_best > best_p                                  # strict: earlier insertions win ties
        best_p = torch.where(better, step_best, best_p)
        cur_seq = beam_seq[:, torch.arange(B, device=dev), first].t()      # [B,L] (rows > t still zero)
        cur_lps = beam_lps[:, torch.arange(B, device=dev), first].t()
        best_seq = torch.where(better.unsqueeze(1), cur_seq, best_seq)
        best_lps = torch.where(better.unsqueeze(1), cur_lps, best_lps)
        best_vix = torch.where(better, first, best_vix)
        sums = torch.where(fin, torch.full_like(sums, -1000.0), sums)
        # next core step with the chosen words (also after the last token, like the reference)
        st = _core_rows(P, st, ops.embed_relu(word.view(-1), P['embed']), fc_gates, pre, pm_rows, K, att2_w)
        att2_ind = att2_w.view(B, K, R).max(dim=2)[1]                # CaptionModelBU.py:182

    att2 = beam_att[:, torch.arange(B, device=dev), best_vix].t().contiguous()   # final column content (view quirk)
    att2[:, 0] = att2_first
    return best_seq, best_lps, att2
