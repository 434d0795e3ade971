"""Seeded synthetic weights and inputs for the GVD hot path.

There is no dataset or checkpoint on the build/GPU boxes, so both the parity tests and bench.py run on
synthetic tensors that honour the reference dataloader's output contract
(/root/reference/misc/dataloader_anet.py:175-354 -> main.py:213-232) and on a
synthetic `state_dict` with the reference's parameter names and shapes, so that the
same weights load into the reference model (oracle/make_golden.py does exactly that) and into ours.

Everything is generated on the CPU with an explicit `torch.Generator`, so the GPU box regenerates
bit-identical tensors from the seed alone (same torch build in both images).
"""
from collections import OrderedDict
import math

import torch


def _uniform(g, shape, bound):
    return (torch.rand(shape, generator=g) * 2.0 - 1.0) * bound


def init_state_dict(opt, seed=0, profile='default'):
    """Synthetic parameters with the reference's `state_dict` keys/shapes (probed).

    Distributions follow torch's default layer inits (Linear/LSTMCell/GRU: U(+-1/sqrt(fan)),
    Embedding: N(0,1)); the Detectron-transferred tensors (model.py:173-211: fc7 -> `ctx2pool_grd`,
    cls_score -> `vis_embed`/`vis_classifiers_bias`) are 0.01*N(0,1) like real fc7/cls_score weights.
    `profile='trained_like'` sharpens the output/attention heads so greedy captions are diverse and
    argmax margins are realistic (random-init captions are degenerate; SURVEY.md §7 hard parts).
    """
    g = torch.Generator().manual_seed(1000003 * (seed + 1))
    H, E, A, V = opt.rnn_size, opt.input_encoding_size, opt.att_hid_size, opt.vocab_size
    D1 = opt.detect_size + 1
    F6, Ffc = opt.att_feat_size, opt.fc_feat_size
    venc = 2048  # transfer_mode in ('none','cls') (model.py:84-85)
    pool_in = F6 + 300 + D1  # model.py:69
    sd = OrderedDict()

    def linear(name, out_f, in_f, bias=True):
        b = 1.0 / math.sqrt(in_f)
        sd[name + '.weight'] = _uniform(g, (out_f, in_f), b)
        if bias:
            sd[name + '.bias'] = _uniform(g, (out_f,), b)

    sd['vis_classifiers_bias'] = 0.01 * torch.randn(D1, generator=g)
    if getattr(opt, 'transfer_mode', 'cls') == 'none':       # (drawn all the same: the other tensors keep their values)
        del sd['vis_classifiers_bias']                         # model.py:198 creates it under 'cls' / 'both' only
    linear('loc_fc.0', 300, 5)
    sd['embed.0.weight'] = torch.randn(V, E, generator=g)
    sd['vis_embed.0.weight'] = 0.01 * torch.randn(D1, venc, generator=g)
    linear('fc_embed.0', H, Ffc + 50)
    linear('seg_info_embed.0', 50, 4)
    linear('att_embed.0.0', H // 2, 2048)
    linear('att_embed.1.0', H // 2, 1024)
    sd['att_embed_aux.0.weight'] = 1.0 + 0.1 * torch.randn(H, generator=g)
    sd['att_embed_aux.0.bias'] = 0.1 * torch.randn(H, generator=g)
    sd['att_embed_aux.0.running_mean'] = 0.1 * torch.randn(H, generator=g)
    sd['att_embed_aux.0.running_var'] = 0.5 + torch.rand(H, generator=g)
    sd['att_embed_aux.0.num_batches_tracked'] = torch.tensor(0, dtype=torch.int64)
    linear('pool_embed.0', H, pool_in)
    linear('ctx2att', A, H)
    linear('ctx2pool', A, H)
    linear('logit', V, H)
    if opt.obj_interact:
        for l in range(2):
            p = 'obj_interact.encoder.layers.%d.' % l
            for w in ('wq', 'wk', 'wv', 'wo'):
                linear(p + 'selfattn.layer.' + w, H, H, bias=False)
            sd[p + 'selfattn.layernorm.gamma'] = 1.0 + 0.05 * torch.randn(H, generator=g)
            sd[p + 'selfattn.layernorm.beta'] = 0.05 * torch.randn(H, generator=g)
            linear(p + 'feedforward.layer.linear1', H // 2, H)
            linear(p + 'feedforward.layer.linear2', H, H // 2)
            sd[p + 'feedforward.layernorm.gamma'] = 1.0 + 0.05 * torch.randn(H, generator=g)
            sd[p + 'feedforward.layernorm.beta'] = 0.05 * torch.randn(H, generator=g)
    hh = H // 2
    ng = 4 if getattr(opt, 't_attn_mode', 'bigru') == 'bilstm' else 3      # gates per unit: nn.LSTM (model.py:145-149) / nn.GRU
    for l in range(2):
        for sfx in ('', '_reverse'):
            b = 1.0 / math.sqrt(hh)
            sd['context_enc.weight_ih_l%d%s' % (l, sfx)] = _uniform(g, (ng * hh, H), b)
            sd['context_enc.weight_hh_l%d%s' % (l, sfx)] = _uniform(g, (ng * hh, hh), b)
            sd['context_enc.bias_ih_l%d%s' % (l, sfx)] = _uniform(g, (ng * hh,), b)
            sd['context_enc.bias_hh_l%d%s' % (l, sfx)] = _uniform(g, (ng * hh,), b)
    sd['ctx2pool_grd.0.weight'] = 0.01 * torch.randn(venc, F6, generator=g)
    sd['ctx2pool_grd.0.bias'] = 0.01 * torch.randn(venc, generator=g)
    b = 1.0 / math.sqrt(H)
    for cell, in_f in (('att_lstm', E + H), ('lang_lstm', 2 * H)):
        sd['core.%s.weight_ih' % cell] = _uniform(g, (4 * H, in_f), b)
        sd['core.%s.weight_hh' % cell] = _uniform(g, (4 * H, H), b)
        sd['core.%s.bias_ih' % cell] = _uniform(g, (4 * H,), b)
        sd['core.%s.bias_hh' % cell] = _uniform(g, (4 * H,), b)
    for att in ('attention', 'attention2'):
        linear('core.%s.h2att' % att, A, H)
        linear('core.%s.alpha_net' % att, 1, A)
    linear('core.i2h_2', H, 2 * H)   # present but unused (AttModel.py:130-131)
    linear('core.h2h_2', H, H)
    if profile == 'trained_like':
        # gains chosen empirically (oracle on CPU): ~50 distinct tokens per 8 captions instead of 1,
        # UNK is top-1 in a few % of steps so the top-2/UNK rule (model.py:590-594) is exercised
        sd['logit.weight'] *= 6.0
        sd['logit.bias'][V - 1] += 2.5
        sd['core.attention2.alpha_net.weight'] *= 8.0
        sd['core.attention.alpha_net.weight'] *= 8.0
        sd['embed.0.weight'] *= 2.0
        sd['core.att_lstm.weight_ih'][:, H:] *= 6.0
        sd['core.att_lstm.weight_hh'] *= 2.0
        sd['core.lang_lstm.weight_ih'] *= 4.0
        sd['core.lang_lstm.weight_ih'][:, H:] *= 2.0
        sd['core.lang_lstm.weight_hh'] *= 1.5
    elif profile != 'default':
        raise ValueError(profile)
    if getattr(opt, 'region_attn_mode', 'mix') == 'dp':      # (drawn all the same: the other tensors keep their values)
        # dot-product region attention: Attention2 has no alpha_net (AttModel.py:63-66); the projections' scale is what sets
        # the sharpness of the softmax there - p . q over 512 terms needs none of the 'trained_like' gain
        del sd['core.attention2.alpha_net.weight'], sd['core.attention2.alpha_net.bias']
    return sd


def make_inputs(opt, batch_size, seed=0, train=True, t_attn_size=None, max_boxes=8, max_cap_len=None, device=None):
    """Synthetic batch with the exact dtypes/shapes main.py hands to `model(...)`.

    Returns a dict of CPU tensors: segs_feat f32[B,Ft,3072], seq i64[B,1,L+1,4], gt_seq i64[B,10,L],
    num i64[B,7], ppls f32[B,R,7], gt_boxes f32[B,NB,6], mask_boxes u8[B,1,NB,L+1],
    ppls_feat f32[B,R,2048], frm_mask u8[B,R,NB], sample_idx i64[B,2], pnt_mask u8[B,R+1].
    With `train=False` the training-only tensors are the `[B]` uint8 dummies main.py:353 passes.

    device (train=False only): generate ON that device with its own generator - same distributions, other values than the
    CPU stream (whose seeds the committed reference cases are keyed on).  For ranks that only need a workload: 8 ranks
    drawing 2.1 GB of host normals each with cpu_count / 8 threads cost minutes of a multi-GPU lease.
    """
    if device is not None and torch.device(device).type != 'cpu':
        assert not train, 'device-side generation covers the inference inputs only'
        dev = torch.device(device)
        g = torch.Generator(device=dev).manual_seed(7919 * (seed + 1) + batch_size)
    else:
        dev = torch.device('cpu')
        g = torch.Generator().manual_seed(7919 * (seed + 1) + batch_size)
    B = batch_size
    T, P = opt.num_sampled_frm, opt.num_prop_per_frm
    R = T * P
    Ft = opt.t_attn_size if t_attn_size is None else t_attn_size
    L, V, D = opt.seq_length, opt.vocab_size, opt.detect_size
    kw = dict(generator=g, device=dev)

    # proposals: x1,y1,x2,y2,frame,cls,score (dataloader_anet.py:188-194); masked rows zeroed (343-344)
    x1 = torch.rand(B, R, **kw) * 500.0
    y1 = torch.rand(B, R, **kw) * 500.0
    w = 5.0 + torch.rand(B, R, **kw) * 200.0
    h = 5.0 + torch.rand(B, R, **kw) * 200.0
    frame = (torch.arange(R, device=dev) // P).float().unsqueeze(0).expand(B, R)
    cls = torch.randint(0, 1601, (B, R), **kw).float()
    score = torch.rand(B, R, **kw)
    ppls = torch.stack([x1, y1, x1 + w, y1 + h, frame, cls, score], dim=2).contiguous()
    ppl_mask = (score <= opt.prop_thresh)
    ppls = ppls.masked_fill(ppl_mask.unsqueeze(-1), 0.0)
    ppls_feat = torch.relu(torch.randn(B, R, opt.att_feat_size, **kw))
    ppls_feat = ppls_feat.masked_fill(ppl_mask.unsqueeze(-1), 0.0)
    pnt_mask = torch.cat([torch.zeros(B, 1, dtype=torch.uint8, device=dev), ppl_mask.to(torch.uint8)], dim=1)

    segs_feat = torch.randn(B, Ft, opt.fc_feat_size, **kw)
    s0 = torch.randint(0, max(Ft // 2, 1), (B,), **kw)
    s1 = s0 + 1 + torch.randint(0, max(Ft - Ft // 2, 1), (B,), **kw)
    sample_idx = torch.stack([s0, s1.clamp(max=Ft)], dim=1).long()

    n_seg = torch.randint(2, 9, (B,), **kw)
    seg_idx = (torch.rand(B, **kw) * n_seg.float()).long()
    nb = torch.randint(3, max_boxes + 1, (B,), **kw) if train else torch.zeros(B, dtype=torch.long, device=dev)
    # main.py:223,572 copy the float `num` into a LongTensor: the two timestamps truncate to 0/1
    num = torch.stack([torch.ones(B, dtype=torch.long, device=dev), torch.full((B,), R, dtype=torch.long, device=dev), nb,
                       seg_idx, n_seg, torch.zeros(B, dtype=torch.long, device=dev),
                       torch.randint(0, 2, (B,), **kw)], dim=1)

    out = dict(segs_feat=segs_feat, num=num, ppls=ppls, ppls_feat=ppls_feat,
               sample_idx=sample_idx, pnt_mask=pnt_mask)
    if not train:
        dummy = torch.zeros(B, dtype=torch.uint8, device=dev)
        out.update(seq=dummy, gt_seq=dummy, gt_boxes=dummy, mask_boxes=dummy, frm_mask=dummy)
        return out

    NB = int(nb.max())
    gt_boxes = torch.zeros(B, NB, 6)
    mask_boxes = torch.ones(B, 1, NB, L + 1, dtype=torch.uint8)
    frm_mask = torch.ones(B, R, NB, dtype=torch.uint8)
    seq = torch.zeros(B, 1, L + 1, 4, dtype=torch.long)
    gt_seq = torch.zeros(B, 10, L, dtype=torch.long)
    for b in range(B):
        cap_len = int(torch.randint(6, (max_cap_len or L) + 1, (1,), generator=g))
        words = torch.randint(1, V - 1, (cap_len,), generator=g)   # never END(0) nor UNK(V-1)
        gt_seq[b, 0, :cap_len] = words
        seq[b, 0, 1:cap_len + 1, 0] = words
        seq[b, 0, 1:cap_len + 1, 3] = words
        nbox = int(nb[b])
        valid = (~ppl_mask[b]).nonzero().view(-1)
        pick = valid[torch.randperm(valid.numel(), generator=g)[:nbox]]
        pos = torch.sort(torch.randperm(cap_len, generator=g)[:nbox])[0]  # boxes sorted by word idx (l.225)
        for k in range(nbox):
            r = int(pick[k])
            jitter = (torch.rand(4, generator=g) - 0.5) * 6.0
            gt_boxes[b, k, :4] = ppls[b, r, :4] + jitter
            gt_boxes[b, k, 4] = ppls[b, r, 4]
            c = int(torch.randint(1, D + 1, (1,), generator=g))
            gt_boxes[b, k, 5] = c
            j = int(pos[k]) if k < pos.numel() else int(pos[-1])
            mask_boxes[b, 0, k, j + 1] = 0                       # dataloader_anet.py:275-277,330
            seq[b, 0, j + 1, 0] = V + c                          # grounded word: V + det idx (l.259)
            seq[b, 0, j + 1, 1] = 1
            seq[b, 0, j + 1, 2] = c
            frm_mask[b, :, k] = (ppls[b, :, 4] != gt_boxes[b, k, 4]).to(torch.uint8)  # l.168-173
        # padded proposals (masked rows have frame 0 after zeroing) keep whatever the comparison gives,
        # exactly like the reference (it compares the already padded/zeroed frame column, l.333)
    out.update(seq=seq, gt_seq=gt_seq, gt_boxes=gt_boxes, mask_boxes=mask_boxes, frm_mask=frm_mask)
    return out


def shard(inputs, rank, world):
    """Slice a batch along dim 0 for rank `rank` of `world` (batch data-parallel)."""
    out = {}
    for k, v in inputs.items():
        n = v.shape[0]
        assert n % world == 0, 'batch %d not divisible by world %d' % (n, world)
        per = n // world
        out[k] = v[rank * per:(rank + 1) * per].contiguous()
    return out


def trim_to_batch(inputs):
    """main.py:213-218 trims proposals/boxes to the batch maxima in `num` before calling the model."""
    out = dict(inputs)
    if inputs['gt_boxes'].dim() == 3:
        nb = max(int(inputs['num'][:, 2].max()), 1)
        out['gt_boxes'] = inputs['gt_boxes'][:, :nb].contiguous()
        out['mask_boxes'] = inputs['mask_boxes'][:, :, :nb].contiguous()
        out['frm_mask'] = inputs['frm_mask'][:, :, :nb].contiguous()
    return out


FORWARD_ORDER = ('segs_feat', 'seq', 'gt_seq', 'num', 'ppls', 'gt_boxes', 'mask_boxes', 'ppls_feat',
                 'frm_mask', 'sample_idx', 'pnt_mask')


def as_args(inputs, device=None):
    """Positional argument tuple in the order of `AttModel.forward` (model.py:227)."""
    return tuple(inputs[k].to(device) if device is not None else inputs[k] for k in FORWARD_ORDER)


def write_feature_split(root, opt, n_segments, segs_per_video=4, seed=0, num_frm=(300, 480, 600)):
    """A synthetic split in the on-disk layout the reference's loader reads (dataloader_anet.py:175-210): per segment
    `fc6_feat_100rois/<vid>_segment_<kk>.npy` f32 [T, P, 2048] (post-ReLU region features), per video
    `rgb_motion_1d/<vid[2:]>_resnet.npy` f32 [F, 2048] and `_bn.npy` f32 [F, 1024]; + the segment records
    ingest.InferenceIngest consumes (seg_id, n_seg_in_vid, timestamps, duration, proposals f64 [T*P, 7] =
    x1, y1, x2, y2, frame, class, score).  -> (feature_root, seg_feature_root, records).  For throughput runs: values are
    random, the files of one video are written once and reused by its segments."""
    import os
    import numpy as np
    rng = np.random.default_rng(seed)
    feature_root = os.path.join(root, 'fc6_feat_100rois')
    seg_root = os.path.join(root, 'rgb_motion_1d')
    os.makedirs(feature_root, exist_ok=True)
    os.makedirs(seg_root, exist_ok=True)
    T, P = opt.num_sampled_frm, opt.num_prop_per_frm
    n = T * P
    records = []
    v = 0
    while len(records) < n_segments:
        vid = 'v_%011d' % (5000 + v)
        F = num_frm[v % len(num_frm)]
        np.save(os.path.join(seg_root, vid[2:] + '_resnet.npy'), rng.standard_normal((F, 2048), dtype=np.float32))
        np.save(os.path.join(seg_root, vid[2:] + '_bn.npy'), rng.standard_normal((F, opt.fc_feat_size - 2048), dtype=np.float32))
        dur = 60.0 + 7.0 * (v % 9)
        nseg = min(segs_per_video, n_segments - len(records))
        for k in range(nseg):
            seg_id = '%s_segment_%02d' % (vid, k)
            feat = np.maximum(rng.standard_normal((T, P, opt.att_feat_size), dtype=np.float32), 0)
            np.save(os.path.join(feature_root, seg_id + '.npy'), feat)
            x1, y1 = rng.random(n) * 500, rng.random(n) * 500
            props = np.stack([x1, y1, x1 + 5 + rng.random(n) * 200, y1 + 5 + rng.random(n) * 200,
                              np.repeat(np.arange(T), P).astype(np.float64), rng.integers(0, 1601, n).astype(np.float64),
                              rng.random(n)], axis=1)
            t0 = dur * k / segs_per_video
            records.append(dict(seg_id=seg_id, n_seg_in_vid=segs_per_video, timestamps=(t0, t0 + 0.9 * dur / segs_per_video),
                                duration=dur, proposals=props))
        v += 1
    return feature_root, seg_root, records
