"""CPU restatement of the reference's dataloader contract (TEST INFRASTRUCTURE): inference inputs and training inputs.

Follows misc/dataloader_anet.py:175-354 (`__getitem__`: region/frame feature files, proposal mask, caption -> input_seq /
gt_seq with grounded words as V + class, GT boxes, box / frame masks, padding, zeroing of masked rows, `num`,
`sample_idx`), torch's default collate, and main.py:213-232 / 339-347 (trim to the batch maxima, legacy pad column of
`pnt_mask`, LongTensor copy of `num`).  PINNED: tests/test_ingest_cpu.py runs the reference's REAL
`DataLoader.__getitem__` (oracle/ref_dataloader_harness.py: stubbed h5py/torchtext/torchvision imports, object built
with __new__) on the same synthetic files and requires identical tensors; tests/golden/ingest_*.npz carries that pin to
the GPU box.  Proposals come as arrays (the reference reads them from `dets_labels` / `dets_num` of an h5 file).

A segment record is a dict: seg_id '<vid>_segment_<k>', n_seg_in_vid, timestamps (t0, t1), duration,
proposals float [n,7] (x1,y1,x2,y2,frame,cls,score) and — for training — caption = the reference's caption-file entry
{'caption': [words], 'clss': [[class names] per box], 'idx': [[word positions] per box], 'bbox': [[x1,y1,x2,y2]],
'frm_idx': [frame per box]} (dataloader_anet.py:212-231).
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


def caption_tensors(rec, vocab, opt, test_mode=False):
    """The training half of `__getitem__` (dataloader_anet.py:212-334) for one segment: input_seq i64 [1,L+1,4],
    gt_seq i64 [10,L], gt_boxes f32 [100,6], box_mask u8 [1,100,L+1], n_box; vocab = dict(wtoi, wtod)."""
    L = opt.seq_length
    cap = rec['caption']
    wtoi, wtod = vocab['wtoi'], vocab['wtod']
    ann = []
    bidx = 0
    for i, clss in enumerate(cap['clss']):                                            # l.216-231
        for j, cls in enumerate(clss):
            if cap['idx'][i][j] < L:
                if test_mode:
                    ann.append(dict(bbox=[0, 0, 0, 0], label=wtod[cls], clss=cls, bbox_idx=bidx, idx=cap['idx'][i][j], frm_idx=-1))
                else:
                    ann.append(dict(bbox=cap['bbox'][i], label=wtod[cls], clss=cls, bbox_idx=bidx, idx=cap['idx'][i][j],
                                    frm_idx=cap['frm_idx'][i]))
                bidx += 1
    ann = sorted(ann, key=lambda x: x['idx'])                                         # l.234 (stable)
    gt = np.zeros((len(ann), 8))
    for i, b in enumerate(ann):
        gt[i, :4], gt[i, 4], gt[i, 5], gt[i, 6], gt[i, 7] = b['bbox'], b['frm_idx'], b['label'], b['bbox_idx'], b['idx']
    if not test_mode:                                                                 # l.244-248: drop degenerate boxes
        gx, gy = gt[:, 2] - gt[:, 0] + 1, gt[:, 3] - gt[:, 1] + 1
        gt = gt[(gx != 1) & (gy != 1)]
    words = cap['caption']
    present = [gt[i, 6] for i in range(gt.shape[0])]                                  # get_det_word, l.128-150
    ind = [(0, 0, 0)] * len(words)
    for b in ann:
        if b['bbox_idx'] in present:
            ind[b['idx']] = (wtod[b['clss']], (b['clss'] != words[b['idx']]) + 1, b['label'])
    cap_seq = np.zeros([1, L, 5])                                                     # l.257-273
    for j in range(min(len(words), L)):
        if ind[j][0] != 0:
            cap_seq[0, j, 0] = ind[j][0] + opt.vocab_size
            cap_seq[0, j, 1], cap_seq[0, j, 2] = ind[j][1], ind[j][2]
            cap_seq[0, j, 3] = wtoi[words[j]]
            cap_seq[0, j, 4] = wtoi[words[j]]
        else:
            cap_seq[0, j, 0] = wtoi[words[j]]
            cap_seq[0, j, 4] = wtoi[words[j]]
    box_mask = np.ones((1, gt.shape[0], L))                                           # l.277-279
    for i in range(gt.shape[0]):
        box_mask[0, i, int(gt[i][7])] = 0
    gt = gt[:, :6]
    input_seq = np.zeros([1, L + 1, 4])
    input_seq[:, 1:] = cap_seq[:, :, :4]                                              # l.296-297 (ncap = seq_per_img = 1)
    gt_seq = np.zeros([10, L])
    gt_seq[:1, :] = cap_seq[:, :, 4]
    n_box = min(gt.shape[0], 100)
    pad_gt = np.zeros((100, 6))
    pad_bm = np.ones((1, 100, L + 1))
    pad_gt[:n_box] = gt[:n_box]
    pad_bm[:, :n_box, 1:] = box_mask[:, :n_box, :]
    return dict(input_seq=torch.from_numpy(input_seq).long(), gt_seq=torch.from_numpy(gt_seq).long(),
                gt_boxes=torch.from_numpy(pad_gt).float(), box_mask=torch.from_numpy(pad_bm).byte(), n_box=n_box)


def assemble_train_batch(records, vocab, feature_root, seg_feature_root, opt, exclude_bgd_det=False):
    """collate + main.py:213-232: the eleven training tensors exactly as `model(..., 'MLE')` receives them."""
    out = assemble_batch(records, feature_root, seg_feature_root, opt, exclude_bgd_det)
    R = opt.num_sampled_frm * opt.num_prop_per_frm
    caps = [caption_tensors(r, vocab, opt) for r in records]
    B = len(records)
    frm = torch.ones(B, R, 100, dtype=torch.uint8)                                    # l.323,333-334
    for b, (r, c) in enumerate(zip(records, caps)):
        n_pps = int(out['num'][b, 1])
        # the frame column of the PADDED proposal buffer before masked rows are zeroed (l.326-333 precede l.343)
        pf = np.zeros(R)
        pf[:n_pps] = np.asarray(r['proposals'], dtype=np.float64)[:n_pps, 4]
        gf = c['gt_boxes'][:c['n_box'], 4].double().numpy()
        frm[b, :n_pps, :c['n_box']] = torch.from_numpy(pf[:n_pps].reshape(-1, 1) != gf.reshape(1, -1)).byte()
        out['num'][b, 2] = c['n_box']
    bmax = max(int(out['num'][:, 2].max()), 1)
    rmax = out['ppls'].shape[1]
    out.update(seq=torch.stack([c['input_seq'] for c in caps]), gt_seq=torch.stack([c['gt_seq'] for c in caps]),
               gt_boxes=torch.stack([c['gt_boxes'] for c in caps])[:, :bmax].contiguous(),
               mask_boxes=torch.stack([c['box_mask'] for c in caps])[:, :, :bmax].contiguous(),
               frm_mask=frm[:, :rmax, :bmax].contiguous())
    return out


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
                                duration=dur, proposals=props, caption=_synthetic_caption(rng, props, opt, len(records))))
    return feature_root, seg_root, records


N_WORDS, N_DET = 60, 12


def synthetic_vocab(opt):
    """wtoi (word -> str index, 1-based like info['ix_to_word'] inverted) and wtod (class name -> 1-based detection
    index, dataloader_anet.py:53) of the synthetic captions."""
    wtoi = {'w%d' % i: str(i) for i in range(1, N_WORDS + 1)}
    wtod = {'w%d' % i: i for i in range(1, N_DET + 1)}            # the first N_DET words are also object classes
    return dict(wtoi=wtoi, wtod=wtod)


def _synthetic_caption(rng, props, opt, serial):
    """A caption-file entry (dataloader_anet.py:212-231): words, boxes with one or two class labels each, some boxes
    past the length limit, one degenerate (zero-area) box, a box whose class name differs from the word it points at."""
    L = opt.seq_length
    n_words = int(rng.randint(5, L + 6))                          # some captions are longer than seq_length
    words = ['w%d' % int(rng.randint(1, N_WORDS + 1)) for _ in range(n_words)]
    nb = int(rng.randint(1, 6))
    pos = sorted(rng.choice(n_words, size=min(nb, n_words), replace=False).tolist(), reverse=bool(serial % 2))
    clss, idx, bbox, frm = [], [], [], []
    for bi, p in enumerate(pos):
        r = int(rng.randint(0, props.shape[0]))
        c = 'w%d' % int(rng.randint(1, N_DET + 1))
        names, where = [c], [p]
        if bi == 0 and n_words > 2:                               # one box with two labels at two word positions
            names.append('w%d' % int(rng.randint(1, N_DET + 1)))
            where.append((p + 1) % n_words)
        if p < L and serial % 3 != 0:
            words[p] = c                                          # the word IS the class name (bn = 1), else bn = 2
        box = (props[r, :4] + rng.randn(4)).tolist()
        if bi == 1 and serial % 2 == 0:
            box = [10.0, 20.0, 10.0, 50.0]                        # x2 - x1 + 1 == 1: dropped by l.244-248
        clss.append(names); idx.append(where); bbox.append(box); frm.append(int(props[r, 4]))
    return dict(caption=words, clss=clss, idx=idx, bbox=bbox, frm_idx=frm)
