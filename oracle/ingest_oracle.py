"""CPU restatement of the reference's feature ingest for the inference inputs (TEST INFRASTRUCTURE).

Follows misc/dataloader_anet.py:175-212,317-354 (`__getitem__`: region/frame feature files, proposal mask, padding,
zeroing of masked rows, `num`, `sample_idx`), torch's default collate, and main.py:339-347 (trim to the batch
maximum, legacy pad column of `pnt_mask`, LongTensor copy of `num`).  PARITY UNPINNED: the reference dataloader cannot
run here (h5py / torchtext / the dataset are absent), so this is a line-by-line restatement, not a
pinned oracle.  Proposals come as arrays (the reference reads them from `dets_labels` / `dets_num` of an h5 file).

A segment record is a dict: seg_id '<vid>_segment_<k>', n_seg_in_vid, timestamps (t0, t1), duration,
proposals float [n,7] (x1,y1,x2,y2,frame,cls,score).
"""
import os

import numpy as np
import torch


def load_segment(rec, feature_root, seg_feature_root, opt, exclude_bgd_det=False):
    """One segment's inference tensors the way dataloader_anet.py:175-212,317-354 builds them (float64 numpy padding
    buffers, byte mask, masked rows zeroed after the float32 conversion)."""
    vid, k = rec['seg_id'].split('_segment_')
    k = int(k)
    boxes = np.array(rec['proposals'], dtype=np.float64, copy=True)                    # [n,7]
    fc6 = np.load(os.path.join(feature_root, rec['seg_id'] + '.npy'))
    fc6 = fc6.reshape(-1, fc6.shape[2]).copy()
    n = boxes.shape[0]
    assert n == fc6.shape[0]                                                           # l.191
    low = boxes[:, 6] <= opt.prop_thresh                                               # l.194
    if exclude_bgd_det:
        low |= boxes[:, 5] == 0                                                        # l.195-196
    frames = np.concatenate((np.load(os.path.join(seg_feature_root, vid[2:] + '_resnet.npy')),
                             np.load(os.path.join(seg_feature_root, vid[2:] + '_bn.npy'))), axis=1)   # l.199-201
    F = frames.shape[0]
    t0, t1 = rec['timestamps']
    dur = rec['duration']
    span = np.array([np.round(F * t0 * 1. / dur), np.round(F * t1 * 1. / dur)])        # l.207 (half-to-even)
    span = np.clip(np.round(span), 0, opt.t_attn_size).astype(int)                     # l.208
    Ft = opt.t_attn_size
    frame_buf = np.zeros((Ft, frames.shape[1]))                                        # l.209-210
    frame_buf[:min(Ft, F)] = frames[:Ft]
    R = opt.num_sampled_frm * opt.num_prop_per_frm                                     # max_proposal, l.45
    keep_n = min(n, R)
    box_buf, mask_buf, feat_buf = np.zeros((R, 7)), np.ones((R)), np.zeros((R, opt.att_feat_size))   # l.318-322
    box_buf[:keep_n], mask_buf[:keep_n], feat_buf[:keep_n] = boxes[:keep_n], low[:keep_n], fc6[:keep_n]
    box_t = torch.from_numpy(box_buf).float()
    mask_t = torch.from_numpy(mask_buf).byte()
    feat_t = torch.from_numpy(feat_buf).float()
    rows = mask_t.bool().view(-1, 1)
    box_t = box_t.masked_fill(rows, 0.)                                                # l.343
    feat_t = feat_t.masked_fill(rows, 0.)                                              # l.344
    num = torch.FloatTensor([1, keep_n, 0, k, rec['n_seg_in_vid'], t0 * 1. / dur, t1 * 1. / dur])   # l.346-348
    return dict(seg_feature=torch.from_numpy(frame_buf), num=num, proposals=box_t, region_feature=feat_t,
                sample_idx=torch.from_numpy(span).long(), pnt_mask=mask_t)


def assemble_batch(records, feature_root, seg_feature_root, opt, exclude_bgd_det=False):
    """collate + main.py:339-347: the six inference tensors exactly as `model(...)` receives them."""
    items = [load_segment(r, feature_root, seg_feature_root, opt, exclude_bgd_det) for r in records]
    seg_feat = torch.stack([it['seg_feature'] for it in items])
    num = torch.stack([it['num'] for it in items])
    proposals = torch.stack([it['proposals'] for it in items])
    region_feat = torch.stack([it['region_feature'] for it in items])
    ppl_mask = torch.stack([it['pnt_mask'] for it in items])
    sample_idx = torch.stack([it['sample_idx'] for it in items])
    rmax = max(int(num[:, 1].max()), 1)
    proposals, ppl_mask, region_feat = proposals[:, :rmax, :], ppl_mask[:, :rmax], region_feat[:, :rmax, :]
    pnt_mask = torch.cat((ppl_mask.new_zeros(ppl_mask.size(0), 1), ppl_mask), dim=1)   # legacy pad column
    return dict(segs_feat=seg_feat.float(), num=num.long(), ppls=proposals.contiguous(),
                ppls_feat=region_feat.contiguous(), sample_idx=sample_idx, pnt_mask=pnt_mask.contiguous())


def write_synthetic_dataset(root, opt, n_videos=3, segs_per_video=(2, 3, 1), seed=0, num_frm=(7, 480, 600),
                            short_props=True):
    """Synthetic feature files + segment records in the on-disk layout the reference reads (feature_root/<seg>.npy
    [T,P,2048] f32, seg_feature_root/<vid[2:]>_resnet.npy [F,2048], _bn.npy [F,1024])."""
    rng = np.random.RandomState(seed)
    feature_root = os.path.join(root, 'fc6_feat_100rois')
    seg_root = os.path.join(root, 'rgb_motion_1d')
    os.makedirs(feature_root, exist_ok=True)
    os.makedirs(seg_root, exist_ok=True)
    T, P = opt.num_sampled_frm, opt.num_prop_per_frm
    records = []
    for v in range(n_videos):
        vid = 'v_%011d' % (1000 + v)
        F = num_frm[v % len(num_frm)]
        np.save(os.path.join(seg_root, vid[2:] + '_resnet.npy'), rng.randn(F, 2048).astype(np.float32))
        np.save(os.path.join(seg_root, vid[2:] + '_bn.npy'), rng.randn(F, opt.fc_feat_size - 2048).astype(np.float32))
        nseg = segs_per_video[v % len(segs_per_video)]
        dur = float(30 + 10 * v)
        for k in range(nseg):
            seg_id = '%s_segment_%02d' % (vid, k)
            # the reference keeps T*P rows per file; one short file exercises the padding path
            t_here = T - 1 if (short_props and v == 1 and k == 0 and T > 1) else T
            feat = np.maximum(rng.randn(t_here, P, opt.att_feat_size), 0).astype(np.float32)
            np.save(os.path.join(feature_root, seg_id + '.npy'), feat)
            n = t_here * P
            x1, y1 = rng.rand(n) * 500, rng.rand(n) * 500
            props = np.stack([x1, y1, x1 + 5 + rng.rand(n) * 200, y1 + 5 + rng.rand(n) * 200,
                              np.repeat(np.arange(t_here), P).astype(np.float64),
                              rng.randint(0, 1601, n).astype(np.float64), rng.rand(n)], axis=1)
            t0 = dur * k / nseg
            records.append(dict(seg_id=seg_id, n_seg_in_vid=nseg, timestamps=(t0, t0 + dur / nseg * 0.9),
                                duration=dur, proposals=props))
    return feature_root, seg_root, records
