"""CPU oracle for the GVD hot path — TEST INFRASTRUCTURE, never the product path.

A from-scratch functional restatement (plain torch-CPU fp32 tensor math over a weight dict; no
nn.Module graph, no reference code) of the algorithm the reference implements in
  misc/model.py      AttModel.forward/_forward/_sample/_sample_beam/_grounder   (227-742)
  misc/AttModel.py   Attention (22-53), Attention2 (56-108), TopDownCore (111-164)
  misc/CaptionModelBU.py  beam_search/beam_step (24-185)
  misc/utils.py      LMCriterion (117-152), bbox_overlaps/sim_mat_target/bbox_target (293-328)
  misc/bbox_transform.py  bbox_overlaps_batch 3-D branch (224-269)
  misc/transformer.py     LayerNorm/Attention/MultiHead/FeedForward/EncoderLayer (66-190)
Each function cites the lines it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module, and only as the checker / the timed CPU baseline.

PINNING.  The reference ships no tests, golden vectors or fixtures: parity is
unpinned by the reference's own tests.  This oracle is instead pinned against the reference *itself*:
tests/test_oracle_vs_reference.py imports the real model from /root/reference (oracle/ref_harness.py)
in the build container and requires identical greedy token ids / attended-region indices and
matching losses on seeded inputs; oracle/make_golden.py stores those reference outputs under
tests/golden/ so the pin travels to the GPU box where /root/reference is absent.
Beam search: the reference's beam path raises as shipped (TypeError at CaptionModelBU.py:179-181,
SURVEY.md §0.4).  `sample_beam` follows the code with the documented minimal repair and is pinned
"reference-with-shim": oracle/ref_harness.beam_shim makes the reference's OWN beam_search run (drops the two
stray core arguments, `.cuda()` -> identity; no reference file edited) and the ids / attended regions must be
identical (tests/test_oracle_vs_reference.py live; tests/golden/beam*.npz for the GPU box).

Mode: evaluation semantics only (dropout = identity, BatchNorm1d uses running stats), which is the only
mode in which CPU and GPU runs are comparable (the RNG streams cannot match).  All tensors fp32.
"""
import math

import torch
import torch.nn.functional as F

MIN_VALUE = -1e8   # AttModel.py:31,66 / model.py:71


# --------------------------------------------------------------------------------------------------
# small building blocks
# --------------------------------------------------------------------------------------------------
def linear(x, W, name):
    return F.linear(x, W[name + '.weight'], W.get(name + '.bias'))


def lstm_cell(x, h, c, W, prefix):
    """nn.LSTMCell (AttModel.py:121,123,139,160): gate order i,f,g,o."""
    gates = F.linear(x, W[prefix + '.weight_ih'], W[prefix + '.bias_ih']) + \
        F.linear(h, W[prefix + '.weight_hh'], W[prefix + '.bias_hh'])
    i, f, g, o = gates.chunk(4, dim=1)
    c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h_new = torch.sigmoid(o) * torch.tanh(c_new)
    return h_new, c_new


def gru_bidir_2layer_loop(x, W, prefix='context_enc'):
    """Explicit 2-layer bidirectional GRU (model.py:150-154,399), gate order r,z,n; eval (no dropout).

    Written as plain time loops so it is a readable spec; `gru_bidir_2layer` is the fast equivalent.
    """
    B, T, _ = x.shape
    inp = x
    for layer in range(2):
        outs = []
        for sfx, order in (('', range(T)), ('_reverse', range(T - 1, -1, -1))):
            w_ih = W['%s.weight_ih_l%d%s' % (prefix, layer, sfx)]
            w_hh = W['%s.weight_hh_l%d%s' % (prefix, layer, sfx)]
            b_ih = W['%s.bias_ih_l%d%s' % (prefix, layer, sfx)]
            b_hh = W['%s.bias_hh_l%d%s' % (prefix, layer, sfx)]
            hid = w_hh.shape[1]
            h = x.new_zeros(B, hid)
            seq = [None] * T
            gi_all = F.linear(inp, w_ih, b_ih)
            for t in order:
                gi = gi_all[:, t]
                gh = F.linear(h, w_hh, b_hh)
                i_r, i_z, i_n = gi.chunk(3, 1)
                h_r, h_z, h_n = gh.chunk(3, 1)
                r = torch.sigmoid(i_r + h_r)
                z = torch.sigmoid(i_z + h_z)
                n = torch.tanh(i_n + r * h_n)
                h = (1.0 - z) * n + z * h
                seq[t] = h
            outs.append(torch.stack(seq, 1))
        inp = torch.cat(outs, 2)
    return inp


_gru_cache = {}


def gru_bidir_2layer(x, W, prefix='context_enc'):
    """Same function through torch's fused CPU GRU (library op, eval mode) — used for speed."""
    key = id(W)
    ent = _gru_cache.get(key)
    hid = W[prefix + '.weight_hh_l0'].shape[1]
    if ent is None or ent[0] is not W:
        gru = torch.nn.GRU(W[prefix + '.weight_ih_l0'].shape[1], hid, 2, dropout=0.0,
                           bidirectional=True, batch_first=True)
        gru.eval()
        _gru_cache.clear()
        _gru_cache[key] = (W, gru)
    else:
        gru = ent[1]
    with torch.no_grad():
        for n, p in gru.named_parameters():
            p.copy_(W[prefix + '.' + n])
    # run outside no_grad so autograd can flow to the input (weights of the GRU are not compared)
    return torch._VF.gru(x, x.new_zeros(4, x.shape[0], hid),
                         [W[prefix + '.' + n] for n, _ in gru.named_parameters()],
                         True, 2, 0.0, False, True, True)[0]


def lstm_bidir_2layer_loop(x, W, prefix='context_enc'):
    """`--t_attn_mode bilstm` (opts.py:60): the explicit 2-layer bidirectional LSTM nn.LSTM(H, H/2, 2, bidirectional,
    batch_first) of model.py:145-149,399 (zero initial state), gate order i,f,g,o; eval (no inter-layer dropout)."""
    B, T, _ = x.shape
    inp = x
    for layer in range(2):
        outs = []
        for sfx, order in (('', range(T)), ('_reverse', range(T - 1, -1, -1))):
            w_ih = W['%s.weight_ih_l%d%s' % (prefix, layer, sfx)]
            w_hh = W['%s.weight_hh_l%d%s' % (prefix, layer, sfx)]
            b_ih = W['%s.bias_ih_l%d%s' % (prefix, layer, sfx)]
            b_hh = W['%s.bias_hh_l%d%s' % (prefix, layer, sfx)]
            hid = w_hh.shape[1]
            h, c = x.new_zeros(B, hid), x.new_zeros(B, hid)
            seq = [None] * T
            gi_all = F.linear(inp, w_ih, b_ih)
            for t in order:
                i, f, g, o = (gi_all[:, t] + F.linear(h, w_hh, b_hh)).chunk(4, 1)
                c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
                h = torch.sigmoid(o) * torch.tanh(c)
                seq[t] = h
            outs.append(torch.stack(seq, 1))
        inp = torch.cat(outs, 2)
    return inp


def lstm_bidir_2layer(x, W, prefix='context_enc'):
    """Same function through torch's fused CPU LSTM (library op, eval mode) - used for speed at Ft = 480."""
    hid = W[prefix + '.weight_hh_l0'].shape[1]
    names = ['%s_l%d%s' % (n, l, sfx) for l in range(2) for sfx in ('', '_reverse')
             for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
    z = x.new_zeros(4, x.shape[0], hid)
    return torch._VF.lstm(x, (z, z), [W[prefix + '.' + n] for n in names], True, 2, 0.0, False, True, True)[0]


def custom_layernorm(x, gamma, beta, eps=1e-6):
    """transformer.py:66-77: unbiased std, eps added to std (not variance)."""
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)
    return gamma * (x - mean) / (std + eps) + beta


def obj_interact(x, W, n_layers=2, n_heads=6, key_bias=None):
    """transformer.py:135-190,244-254 as configured at model.py:126-135 (no mask, no pos-enc, eval).
    key_bias (NOT in the reference; [B,R] added to every query's scores of a key: log n = the key stands for n identical
    keys, -inf = the key does not exist): lets tests/test_train_compact_cpu.py restate the compacted training layout."""
    d_model = x.shape[-1]
    scale = math.sqrt(d_model)   # Attention(d_key=d_model): transformer.py:92,112
    for l in range(n_layers):
        p = 'obj_interact.encoder.layers.%d.' % l
        q = F.linear(x, W[p + 'selfattn.layer.wq.weight'])
        k = F.linear(x, W[p + 'selfattn.layer.wk.weight'])
        v = F.linear(x, W[p + 'selfattn.layer.wv.weight'])
        heads = []
        for qh, kh, vh in zip(q.chunk(n_heads, -1), k.chunk(n_heads, -1), v.chunk(n_heads, -1)):
            dots = torch.matmul(qh, kh.transpose(1, 2)) / scale
            if key_bias is not None:
                dots = dots + key_bias.unsqueeze(1)
            heads.append(torch.matmul(F.softmax(dots, dim=-1), vh))
        att = F.linear(torch.cat(heads, -1), W[p + 'selfattn.layer.wo.weight'])
        x = custom_layernorm(x + att, W[p + 'selfattn.layernorm.gamma'], W[p + 'selfattn.layernorm.beta'])
        ff = linear(F.relu(linear(x, W, p + 'feedforward.layer.linear1')), W, p + 'feedforward.layer.linear2')
        x = custom_layernorm(x + ff, W[p + 'feedforward.layernorm.gamma'], W[p + 'feedforward.layernorm.beta'])
    return x


def grounder_dot(xt, att_feats, mask, bias):
    """model.py:243-280, dot-product branch (no alpha_net on AttModel under 'mix', model.py:55-58)."""
    dot = torch.matmul(xt, att_feats.permute(0, 2, 1).contiguous())
    if bias is not None:
        dot = dot + bias
    m = mask.bool()
    if m.dim() == 2:
        m = m.unsqueeze(1).expand_as(dot)
    return dot.masked_fill(m, MIN_VALUE)


# --------------------------------------------------------------------------------------------------
# box targets (integer / index work: must be bit-exact)
# --------------------------------------------------------------------------------------------------
def bbox_overlaps(ppls, gt_boxes, frm_mask):
    """utils.py:293-297 -> bbox_transform.py:224-269 (3-D branch, '+1' pixel convention).

    ppls [B,R,>=5], gt_boxes [B,K,>=5], frm_mask u8/bool [B,R,K] (1 = different frame or masked).
    """
    a = ppls[:, :, :5]
    g = gt_boxes[:, :, :5]
    B, N, K = a.shape[0], a.shape[1], g.shape[1]
    gx = g[:, :, 2] - g[:, :, 0] + 1
    gy = g[:, :, 3] - g[:, :, 1] + 1
    g_area = (gx * gy).view(B, 1, K)
    ax = a[:, :, 2] - a[:, :, 0] + 1
    ay = a[:, :, 3] - a[:, :, 1] + 1
    a_area = (ax * ay).view(B, N, 1)
    g_zero = (gx == 1) & (gy == 1)
    a_zero = (ax == 1) & (ay == 1)
    bx = a.view(B, N, 1, 5)
    qx = g.view(B, 1, K, 5)
    iw = (torch.min(bx[..., 2], qx[..., 2]) - torch.max(bx[..., 0], qx[..., 0]) + 1).clamp(min=0)
    ih = (torch.min(bx[..., 3], qx[..., 3]) - torch.max(bx[..., 1], qx[..., 1]) + 1).clamp(min=0)
    ua = a_area + g_area - iw * ih
    ov = iw * ih / ua
    ov = ov * (1 - frm_mask.to(torch.uint8)).to(ov.dtype)
    ov = ov.masked_fill(g_zero.view(B, 1, K).expand(B, N, K), 0)
    ov = ov.masked_fill(a_zero.view(B, N, 1).expand(B, N, K), -1)
    return ov


def sim_mat_target(overlaps, box_cls):
    """utils.py:299-305: [B,K,R] class label where IoU > 0.5 else 0."""
    B, N, K = overlaps.shape
    lab = (overlaps > 0.5).long() * box_cls.view(B, 1, K).long()
    return lab.permute(0, 2, 1).contiguous()


def roi_labels_for_step(mask_boxes_t, overlaps):
    """utils.py:307-328 (`bbox_target`; its seq_update side effect is dead code): [B,R] float 0/1.

    mask_boxes_t u8 [B,1,K]: 0 at the box tied to this word position.
    """
    B = overlaps.shape[0]
    ov = overlaps.masked_fill(mask_boxes_t.reshape(B, 1, -1).bool().expand_as(overlaps), 0)
    return (ov.max(dim=2)[0] > 0.5).to(ov.dtype)          # (.float() in the reference; dtype-generic for the fp64 truth runs of the tests)


def frame_mask_for_step(mask_boxes_t, frm_mask, pnt_mask):
    """model.py:436-440: [B,R+1] u8; 1 = proposal not on the frame of any box tied to this word."""
    B, R, K = frm_mask.shape
    box_mask = mask_boxes_t.reshape(B, 1, K).expand(B, R, K).to(torch.uint8)
    on = (1 - (box_mask | frm_mask.to(torch.uint8))).sum(dim=2) <= 0
    on = torch.cat([torch.zeros(B, 1, dtype=torch.bool), on], dim=1)
    return (on | pnt_mask.bool()).to(torch.uint8)


# --------------------------------------------------------------------------------------------------
# per-segment preamble  (model.py:302-409 / 504-568 / 634-698 — identical in the three drivers)
# --------------------------------------------------------------------------------------------------
def preamble(W, opt, segs_feat, num, ppls, ppls_feat, sample_idx, pnt_mask, fast_gru=True, bn_train=False,
             enc_key_bias=None):
    B, Ft = segs_feat.shape[0], segs_feat.shape[1]
    T = opt.num_sampled_frm
    D1 = opt.detect_size + 1
    out = {}
    # fc feature: temporal mean ‖ segment-position embedding, each layer-normed (model.py:306-308)
    fc = segs_feat.mean(dim=1)
    seg_info = F.relu(linear(num[:, 3:7].to(W['seg_info_embed.0.weight'].dtype), W, 'seg_info_embed.0'))
    fc = torch.cat([F.layer_norm(fc, [fc.shape[-1]]), F.layer_norm(seg_info, [seg_info.shape[-1]])], dim=-1)
    # fc7 on the raw fc6 region features (model.py:311-313)
    g_pool = F.relu(linear(ppls_feat, W, 'ctx2pool_grd.0'))
    out['g_pool'] = g_pool
    # region-class similarity (model.py:321-340)
    vis_word = F.relu(W['vis_embed.0.weight'])                       # embedding of 0..D1-1, then ReLU
    p_vis = vis_word.view(1, D1, -1).expand(B, D1, vis_word.shape[1]).contiguous()
    # (transfer_mode='none': the reference has no `vis_classifiers_bias` and passes bias = None, model.py:328-332)
    bias = W['vis_classifiers_bias'].view(1, -1, 1).expand(B, D1, g_pool.shape[1]) if 'vis_classifiers_bias' in W else None
    sim_logits = grounder_dot(p_vis, g_pool, pnt_mask[:, 1:], bias)
    sim_mat = F.softmax(sim_logits, dim=1)
    out['sim_mat_static'] = sim_mat
    # location + class-distribution features (model.py:357-364)
    loc_in = torch.cat([ppls[:, :, :4] / 720., (ppls[:, :, 4] * 1. / T).unsqueeze(-1)], dim=2)
    loc = F.relu(linear(loc_in, W, 'loc_fc.0'))
    label = sim_mat.permute(0, 2, 1).contiguous()
    pool = torch.cat([F.layer_norm(g_pool, [g_pool.shape[-1]]), F.layer_norm(loc, [loc.shape[-1]]),
                      F.layer_norm(label, [label.shape[-1]])], dim=2)
    fc = F.relu(linear(fc, W, 'fc_embed.0'))                          # model.py:383
    pool = F.relu(linear(pool, W, 'pool_embed.0'))                    # model.py:384
    if opt.obj_interact:
        pool = obj_interact(pool, W, key_bias=enc_key_bias)           # model.py:387-388
    p_pool = linear(pool, W, 'ctx2pool')                              # model.py:391
    out['att_input_mode'] = getattr(opt, 'att_input_mode', 'both')
    out['region_attn_mode'] = getattr(opt, 'region_attn_mode', 'mix')
    if out['att_input_mode'] == 'region':
        # model.py:406-409: no frame-wise context at all (the reference passes 1 x 1 dummies the core never reads)
        out.update(fc=fc, pool=pool, p_pool=p_pool, conv=None, p_conv=None)
        return out
    # frame-wise context (model.py:393-405)
    c = torch.cat([F.relu(linear(segs_feat[:, :, :2048], W, 'att_embed.0.0')),
                   F.relu(linear(segs_feat[:, :, 2048:], W, 'att_embed.1.0'))], dim=2)
    # nn.BatchNorm1d(H) over [B,H,Ft] (model.py:114,397).  eval: running statistics; train (`bn_train`, the dropout-free
    # train-mode parity case): batch statistics over (B, Ft) and the momentum-0.1 update of CLONES of the running buffers
    # (returned as out['bn_running'])
    rm, rv = W['att_embed_aux.0.running_mean'], W['att_embed_aux.0.running_var']
    if bn_train:
        rm, rv = rm.clone(), rv.clone()
        out['bn_running'] = (rm, rv)
    c = F.batch_norm(c.permute(0, 2, 1).contiguous(), rm, rv, W['att_embed_aux.0.weight'],
                     W['att_embed_aux.0.bias'], bool(bn_train), 0.1, 1e-5)
    c = F.relu(c).permute(0, 2, 1).contiguous()
    if getattr(opt, 't_attn_mode', 'bigru') == 'bilstm':               # model.py:145-149
        c = lstm_bidir_2layer(c, W) if fast_gru else lstm_bidir_2layer_loop(c, W)
    else:
        c = gru_bidir_2layer(c, W) if fast_gru else gru_bidir_2layer_loop(c, W)
    idx_mask = torch.ones(B, Ft, 1, dtype=torch.bool)
    for b in range(B):
        idx_mask[b, int(sample_idx[b, 0]):int(sample_idx[b, 1])] = False     # model.py:303-305
    conv = c.masked_fill(idx_mask, 0)
    p_conv = linear(conv, W, 'ctx2att')
    out.update(fc=fc, pool=pool, p_pool=p_pool, conv=conv, p_conv=p_conv)
    return out


# --------------------------------------------------------------------------------------------------
# the per-token step  (AttModel.py:134-164)
# --------------------------------------------------------------------------------------------------
def attention_temporal(h, conv, p_conv, W):
    """AttModel.py:33-53: additive attention over Ft positions, no mask."""
    q = linear(h, W, 'core.attention.h2att')
    dot = torch.tanh(p_conv + q.unsqueeze(1))
    e = linear(dot, W, 'core.attention.alpha_net').squeeze(-1)
    w = F.softmax(e, dim=1)
    return torch.bmm(w.unsqueeze(1), conv).squeeze(1)


def attention_region(h, pool, p_pool, att_mask, pnt_mask, W, mode='mix'):
    """AttModel.py:71-108: returns (context, masked pre-softmax logits, q).  region_attn_mode 'mix' -> additive score
    alpha_net(tanh(p + q)) (AttModel.py:84-91), 'mix_mul' -> alpha_net(tanh(p * q)) (AttModel.py:82-83), 'dp' -> the plain
    dot product p . q, no alpha_net in the module (AttModel.py:92-95)."""
    q = linear(h, W, 'core.attention2.h2att')
    if mode == 'dp':
        e = torch.matmul(p_pool, q.unsqueeze(-1)).squeeze(-1)
    else:
        dot = torch.tanh(p_pool * q.unsqueeze(1) if mode == 'mix_mul' else p_pool + q.unsqueeze(1))
        e = linear(dot, W, 'core.attention2.alpha_net').squeeze(-1)
    e = e.masked_fill(att_mask.bool(), MIN_VALUE)
    w = F.softmax(e, dim=1)
    logits = e.masked_fill(pnt_mask.bool(), MIN_VALUE)
    ctx = torch.bmm(w.unsqueeze(1), pool).squeeze(1)
    return ctx, logits, q


def core_step(W, xt, pre, att_mask, pnt_mask, state):
    """TopDownCore.forward (AttModel.py:134-164). state = (h[2,B,H], c[2,B,H]).  att_input_mode (AttModel.py:140-151):
    'both' feeds att + att2 to the language LSTM, 'featmap' the frame-wise context alone (the region attention still runs:
    its logits are the grounding output), 'region' the region context alone (no frame-wise attention)."""
    h, c = state
    mode = pre.get('att_input_mode', 'both')
    h_att, c_att = lstm_cell(torch.cat([pre['fc'], xt], 1), h[0], c[0], W, 'core.att_lstm')
    att = attention_temporal(h_att, pre['conv'], pre['p_conv'], W) if mode != 'region' else None
    att2, att2_logits, att_h = attention_region(h_att, pre['pool'], pre['p_pool'],
                                                att_mask[:, 1:], pnt_mask[:, 1:], W, pre.get('region_attn_mode', 'mix'))
    ctx = att + att2 if mode == 'both' else (att if mode == 'featmap' else att2)
    h_lang, c_lang = lstm_cell(torch.cat([ctx, h_att], 1), h[1], c[1], W, 'core.lang_lstm')
    return h_lang, (torch.stack([h_att, h_lang]), torch.stack([c_att, c_lang])), att2_logits, att_h


def embed_word(W, it):
    return F.relu(F.embedding(it, W['embed.0.weight']))              # model.py:79-82 (dropout eval)


def word_logprobs(W, h_lang):
    return F.log_softmax(linear(h_lang, W, 'logit'), dim=-1)          # model.py:464,615 (beta = 1)


# --------------------------------------------------------------------------------------------------
# drivers
# --------------------------------------------------------------------------------------------------
def sample_greedy(W, opt, segs_feat, num, ppls, ppls_feat, sample_idx, pnt_mask, pre=None):
    """AttModel._sample with sample_max=1, beam_size=1 (model.py:492-624).

    Returns seq i64[B,L], seqLogprobs f32[B,L], att2_weights f32[B,L,R] (masked logits), sim_mat f32[B,D1,R].
    """
    if pre is None:
        pre = preamble(W, opt, segs_feat, num, ppls, ppls_feat, sample_idx, pnt_mask)
    B, H, L = segs_feat.shape[0], opt.rnn_size, opt.seq_length
    unk = int(opt.wtoi['UNK'])
    state = (torch.zeros(2, B, H), torch.zeros(2, B, H))
    seq, lps, atts = [], [], []
    logprobs = None
    for t in range(L + 1):
        if t == 0:
            it = torch.zeros(B, dtype=torch.long)                                 # BOS (model.py:588)
        else:
            v, i = torch.topk(logprobs, 2, dim=1)                                 # model.py:590-594
            keep = i[:, 0] != unk
            it = torch.where(keep, i[:, 0], i[:, 1])
            lp = torch.where(keep, v[:, 0], v[:, 1])
            seq.append(it)
            lps.append(lp)
        if t < L:
            out, state, att2_logits, _ = core_step(W, embed_word(W, it), pre, pnt_mask, pnt_mask, state)
            logprobs = word_logprobs(W, out)
            atts.append(att2_logits)
    return torch.stack(seq, 1), torch.stack(lps, 1), torch.stack(atts, 1), pre['sim_mat_static']


def attended_region_indices(att2_weights, opt):
    """main.py:364-365: per-frame argmax over the P proposals of each of the T frames -> [B,L,T]."""
    B, L = att2_weights.shape[0], att2_weights.shape[1]
    return att2_weights.view(B, L, opt.num_sampled_frm, opt.num_prop_per_frm).max(dim=-1)[1]


def lm_criterion(logp, att2_weights, ground_weights, target, att2_target):
    """utils.py:122-152. logp [B*Lc,V]; *_weights [B,Lc,R]; target i64 [B,Lc]; att2_target f32 0/1."""
    mask = target > 0
    mask = torch.cat([torch.ones(mask.shape[0], 1, dtype=torch.bool), mask[:, :-1]], 1)
    sel = torch.gather(logp, 1, target.reshape(-1, 1))
    lm = (-torch.masked_select(sel, mask.reshape(-1, 1))).mean()
    tgt = att2_target.bool()
    att2 = -torch.masked_select(F.log_softmax(att2_weights, dim=2), tgt).mean()
    grd = -torch.masked_select(F.log_softmax(ground_weights, dim=2), tgt).mean()
    return lm, att2, grd


def forward_train(W, opt, segs_feat, input_seq, gt_seq, num, ppls, gt_boxes, mask_boxes, ppls_feat,
                  frm_mask, sample_idx, pnt_mask, eval_obj_ground=False, pre=None, bn_train=False):
    """AttModel._forward (model.py:283-489), seq_per_img = 1.

    'MLE' (eval_obj_ground=False) -> (lm_loss, att2_loss, ground_loss, cls_loss) scalars (+ aux dict);
    'GRD' (True) -> (cls_pred i64[N,2], att2_ind i64[B,Lc,T], grd_ind i64[B,Lc,T]).
    """
    assert opt.seq_per_img == 1
    B, R = segs_feat.shape[0], ppls.shape[1]
    H, L, V = opt.rnn_size, opt.seq_length, opt.vocab_size
    seq = gt_seq[:, 0, :]
    seq = torch.cat([torch.zeros(B, 1, dtype=seq.dtype), seq], 1)                     # model.py:285-286
    input_seq = input_seq.view(-1, input_seq.shape[2], input_seq.shape[3])            # [B,L+1,4]
    if pre is None:
        pre = preamble(W, opt, segs_feat, num, ppls, ppls_feat, sample_idx, pnt_mask, bn_train=bn_train)
    pm = pnt_mask.to(torch.uint8)
    fm = frm_mask.to(torch.uint8)
    overlaps = bbox_overlaps(ppls, gt_boxes, fm | pm[:, 1:].unsqueeze(-1))            # model.py:317-318
    sim_target = sim_mat_target(overlaps, gt_boxes[:, :, 5])                          # model.py:345
    sim_mask = sim_target > 0
    sim_mat = pre['sim_mat_static']
    if not eval_obj_ground:
        p = torch.masked_select(torch.gather(sim_mat, 1, sim_target), sim_mask)       # model.py:348-350
        cls_loss = F.binary_cross_entropy(p, torch.ones_like(p))
        cls_pred = None
    else:
        tgt = torch.masked_select(sim_target, sim_mask)                               # model.py:353-355
        prd = torch.masked_select(sim_mat.max(dim=1)[1].unsqueeze(1).expand_as(sim_target), sim_mask)
        cls_pred = torch.stack([tgt, prd], dim=1)
        cls_loss = None

    state = (torch.zeros(2, B, H), torch.zeros(2, B, H))
    outs, atts, labels, fmasks = [], [], [], []
    for i in range(L):                                                                # model.py:421-453
        if i >= 1 and int(seq[:, i].sum()) == 0:
            break
        xt = embed_word(W, seq[:, i])
        if not eval_obj_ground:
            labels.append(roi_labels_for_step(mask_boxes[:, :, :, i + 1], overlaps))
            fmask = frame_mask_for_step(mask_boxes[:, 0, :, i + 1], fm, pm)
            fmasks.append(fmask)
            out, state, a2, _ = core_step(W, xt, pre, pm, fmask, state)
        else:
            out, state, a2, _ = core_step(W, xt, pre, pm, pm, state)
        outs.append(out)
        atts.append(a2)
    Lc = len(outs)
    rnn_out = torch.stack(outs, 1)
    att2_weights = torch.stack(atts, 1)
    logp = word_logprobs(W, rnn_out).view(Lc * B, -1)                                 # model.py:464-465
    xt_clamp = torch.clamp(input_seq[:, 1:Lc + 1, 0] - V, min=0)                      # model.py:469
    xt_all = F.relu(F.embedding(xt_clamp, W['vis_embed.0.weight']))
    # (transfer_mode='none': bias = 0, model.py:472-476)
    bias = W['vis_classifiers_bias'][xt_clamp].unsqueeze(2).expand(B, Lc, R) if 'vis_classifiers_bias' in W else 0
    if not eval_obj_ground:
        fmask_all = torch.stack(fmasks, 1)
        ground = grounder_dot(xt_all, pre['g_pool'], fmask_all[:, :, 1:], bias + att2_weights)
        roi = torch.stack(labels, 1)
        lm, a2l, gl = lm_criterion(logp, att2_weights, ground, seq[:, 1:Lc + 1], roi)
        aux = dict(att2_weights=att2_weights, ground_weights=ground, roi_labels=roi, frm_masks=fmask_all,
                   overlaps=overlaps, sim_target=sim_target, logp=logp, seq_cnt=Lc, bn_running=pre.get('bn_running'))
        return lm, a2l, gl, cls_loss, aux
    ground = grounder_dot(xt_all, pre['g_pool'], pm[:, 1:], bias + att2_weights)
    T, P = opt.num_sampled_frm, opt.num_prop_per_frm
    return (cls_pred, att2_weights.view(B, Lc, T, P).max(dim=-1)[1],
            ground.view(B, Lc, T, P).max(dim=-1)[1])


def sample_beam(W, opt, segs_feat, num, ppls, ppls_feat, sample_idx, pnt_mask, beam_size=5, pre=None):
    """AttModel._sample_beam + CaptionModel.beam_search (model.py:627-742; CaptionModelBU.py:24-185)
    with the minimal repair of SURVEY.md §3.4: the core is called with its 10 real arguments
    (CaptionModelBU.py:179-181 passes 12 -> TypeError in the reference), no `.cuda()`, and
    `sim_mat_static` is returned as the 4th value.  Pinned against the reference's own beam_search run under
    oracle/ref_harness.beam_shim ("reference-with-shim").

    Reproduced quirks: candidates are ordered word-rank-major / beam-minor and stably sorted by
    descending summed log-prob (CaptionModelBU.py:49-61); at t=0 only beam 0 is expanded (48-49);
    no UNK suppression (130-131); finished beams get sum=-1000 but stay in the pool (166); the
    pointer-mask update is a no-op for word tokens (151-152,169-175); the core also runs after the last
    token; att2 of a finished beam is a *view* of column `vix` of beam_att2_ind (159, no clone), so it
    reflects that column's content at the END of the search; att2 holds global argmax-over-R indices.
    Returns seq i64[B,L], seqLogprobs f32[B,L], att2 i64[B,L], sim_mat.
    """
    if pre is None:
        pre = preamble(W, opt, segs_feat, num, ppls, ppls_feat, sample_idx, pnt_mask)
    B, H, L = segs_feat.shape[0], opt.rnn_size, opt.seq_length
    K = beam_size
    seq_out = torch.zeros(L, B, dtype=torch.long)
    lp_out = torch.zeros(L, B)
    att2_out = torch.full((L, B), -1, dtype=torch.long)
    for k in range(B):
        pk = {n: (None if pre[n] is None else pre[n][k:k + 1].expand(K, *pre[n].shape[1:]).contiguous())
              for n in ('fc', 'pool', 'p_pool', 'conv', 'p_conv')}
        pk.update(att_input_mode=pre.get('att_input_mode', 'both'), region_attn_mode=pre.get('region_attn_mode', 'mix'))
        pmk = pnt_mask[k:k + 1].expand(K, pnt_mask.shape[1]).contiguous()
        state = (torch.zeros(2, K, H), torch.zeros(2, K, H))
        rnn_out, state, a2, _ = core_step(W, embed_word(W, torch.zeros(K, dtype=torch.long)), pk, pmk, pmk, state)
        att2_out[0, k] = a2.max(dim=1)[1][0]                                        # model.py:733
        beam_seq = torch.zeros(L, K, dtype=torch.long)
        beam_lps = torch.zeros(L, K)
        beam_att = torch.full((L, K), -1, dtype=torch.long)
        att2_ind = torch.full((K,), -1, dtype=torch.long)
        sums = torch.zeros(K)
        done = []     # (p, seq clone, logps clone, vix)
        for t in range(L):
            lpf = word_logprobs(W, rnn_out)
            ys, ix = torch.sort(lpf, 1, True)
            rows = 1 if t == 0 else K
            cands = []
            for c in range(min(K, ys.shape[1])):
                for q in range(rows):
                    cands.append((sums[q] + ys[q, c], int(ix[q, c]), q, ys[q, c].clone()))   # fp32 sum
            # stable sort by descending p; p compared in fp32 like the reference's tensor scalars
            order = sorted(range(len(cands)), key=lambda j: -float(cands[j][0]))
            new_h, new_c, new_out = state[0].clone(), state[1].clone(), rnn_out.clone()
            prev_seq, prev_lps, prev_att = beam_seq[:t].clone(), beam_lps[:t].clone(), beam_att[:t].clone()
            new_sums = sums.clone()
            for vix in range(K):
                p, cw, q, r = cands[order[vix]]
                if t >= 1:
                    beam_seq[:t, vix] = prev_seq[:, q]
                    beam_lps[:t, vix] = prev_lps[:, q]
                    beam_att[:t, vix] = prev_att[:, q]
                new_h[:, vix] = state[0][:, q]
                new_c[:, vix] = state[1][:, q]
                new_out[vix] = rnn_out[q]
                beam_seq[t, vix] = cw
                beam_lps[t, vix] = r
                if t >= 1:
                    beam_att[t, vix] = att2_ind[q]
                new_sums[vix] = p
            sums = new_sums
            state, rnn_out = (new_h, new_c), new_out
            it = beam_seq[t].clone()
            for vix in range(K):
                if int(beam_seq[t, vix]) == 0 or t == L - 1:
                    done.append((float(sums[vix]), beam_seq[:, vix].clone(), beam_lps[:, vix].clone(), vix))
                    sums[vix] = -1000.0
            rnn_out, state, a2, _ = core_step(W, embed_word(W, it), pk, pmk, pmk, state)
            att2_ind = a2.max(dim=1)[1]
        best = sorted(range(len(done)), key=lambda j: -done[j][0])[0]
        p, s, l, vix = done[best]
        seq_out[:, k] = s
        lp_out[:, k] = l
        att2_out[1:, k] = beam_att[1:, vix]           # the aliasing quirk: final content of column vix
    return seq_out.t().contiguous(), lp_out.t().contiguous(), att2_out.t().contiguous(), pre['sim_mat_static']
