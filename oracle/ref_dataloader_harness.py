"""Run the REAL reference dataloader `misc.dataloader_anet.DataLoader.__getitem__` (dataloader_anet.py:175-354) on
synthetic files (build container only) — the pin for oracle/ingest_oracle.py.  TEST INFRASTRUCTURE.

The module imports h5py / torchtext / torchvision at the top (absent here) and its constructor reads GloVe and the
dataset json/h5 files; `__getitem__` itself needs none of them.  So, without editing any reference file:
  * empty stub modules are placed in sys.modules for exactly those imports;
  * the object is made with `DataLoader.__new__` and given the attributes `__getitem__` reads (the same ones the
    constructor would derive from the dataset files, dataloader_anet.py:27-127), built from our synthetic records.
"""
import os
import sys
import types
from collections import defaultdict

import numpy as np
import torch

from . import ref_harness

_mod = None


def reference_dataloader_module():
    global _mod
    if _mod is None:
        assert ref_harness.reference_available()
        ref_harness._install_uint8_mask_shim()        # PyTorch-1.1 byte-mask semantics (masked_fill_ at l.343-344)
        for name in ('h5py', 'torchtext', 'torchtext.vocab', 'torchvision', 'torchvision.datasets',
                     'torchvision.datasets.folder', 'torchvision.transforms'):
            if name not in sys.modules:
                sys.modules[name] = types.ModuleType(name)
        sys.modules['torchvision.datasets.folder'].default_loader = None
        sys.modules['torchtext'].vocab = sys.modules['torchtext.vocab']
        sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
        sys.modules['torchvision'].datasets = sys.modules['torchvision.datasets']
        sys.modules['torchvision.datasets'].folder = sys.modules['torchvision.datasets.folder']
        if ref_harness.REFERENCE_ROOT not in sys.path:
            sys.path.insert(0, ref_harness.REFERENCE_ROOT)
        from misc import dataloader_anet
        _mod = dataloader_anet
    return _mod


def build_reference_dataset(records, vocab, feature_root, seg_feature_root, opt, exclude_bgd_det=False, test_mode=False):
    """A reference DataLoader object over `records` (oracle/ingest_oracle.py schema, with 'caption' entries) and
    `vocab` = dict(wtoi, wtod): every attribute `__getitem__` reads, derived the way the constructor derives them."""
    DL = reference_dataloader_module().DataLoader
    d = DL.__new__(DL)
    d.seq_per_img = 1
    d.seq_length = opt.seq_length
    d.att_feat_size = opt.att_feat_size
    d.vis_attn = False
    d.feature_root, d.seg_feature_root = feature_root, seg_feature_root
    d.num_sampled_frm, d.num_prop_per_frm = opt.num_sampled_frm, opt.num_prop_per_frm
    d.exclude_bgd_det = exclude_bgd_det
    d.prop_thresh = opt.prop_thresh
    d.t_attn_size = opt.t_attn_size
    d.test_mode = test_mode
    d.max_gt_box = 100
    d.max_proposal = d.num_sampled_frm * d.num_prop_per_frm
    d.wtoi = dict(vocab['wtoi'])
    d.wtod = dict(vocab['wtod'])           # word -> detection index (1-based), dataloader_anet.py:53
    d.dtoi = d.wtod
    d.vocab_size = opt.vocab_size
    d.info = {'videos': [{'id': r['seg_id']} for r in records]}
    d.split_ix = list(range(len(records)))
    d.num_seg_per_vid = defaultdict(list)
    for r in records:
        vid, k = r['seg_id'].split('_segment_')
        d.num_seg_per_vid[vid] = list(range(r['n_seg_in_vid']))          # max()+1 == n_seg_in_vid
    R = d.max_proposal
    d.num_proposals = np.array([r['proposals'].shape[0] for r in records])
    d.label_proposals = np.zeros((len(records), max(R, int(d.num_proposals.max())), 7))
    for i, r in enumerate(records):
        d.label_proposals[i, :r['proposals'].shape[0]] = r['proposals']
    d.timestamp_file = {'annotations': defaultdict(lambda: {'segments': {}})}
    d.caption_file = defaultdict(lambda: {'segments': {}})
    for r in records:
        vid, k = r['seg_id'].split('_segment_')
        k = str(int(k))
        d.timestamp_file['annotations'][vid]['duration'] = r['duration']
        d.timestamp_file['annotations'][vid]['segments'][k] = {'timestamps': list(r['timestamps'])}
        d.caption_file[vid]['segments'][k] = r['caption']
    return d


def reference_batch(dataset, indices, train):
    """default collate of `dataset[i]` + the trimming main.py applies before `model(...)` (main.py:213-232 for train,
    339-347 for eval) -> dict of the model inputs."""
    items = [dataset[i] for i in indices]
    seg_feat = torch.stack([torch.as_tensor(it[0]) for it in items])
    iseq, gts, num, proposals, bboxs, box_mask = (torch.stack([it[j] for it in items]) for j in range(1, 7))
    seg_ids = [it[7] for it in items]
    region_feat, frm_mask, sample_idx, ppl_mask = (torch.stack([it[j] for it in items]) for j in range(8, 12))
    rmax = max(int(max(num[:, 1])), 1)
    proposals, ppl_mask, region_feat = proposals[:, :rmax, :], ppl_mask[:, :rmax], region_feat[:, :rmax, :]
    out = dict(seg_ids=seg_ids, segs_feat=seg_feat.float(), num=num.long(), ppls=proposals.contiguous(),
               ppls_feat=region_feat.contiguous(), sample_idx=sample_idx,
               pnt_mask=torch.cat((ppl_mask.new_zeros(ppl_mask.size(0), 1), ppl_mask), dim=1).contiguous())
    if train:
        bmax = max(int(max(num[:, 2])), 1)
        out.update(seq=iseq, gt_seq=gts, gt_boxes=bboxs[:, :bmax, :].contiguous(),
                   mask_boxes=box_mask[:, :, :bmax, :].contiguous(), frm_mask=frm_mask[:, :rmax, :bmax].contiguous())
    return out
