"""Host half of the feature ingest (grounded-video-description_amd/ingest.py) against the restated reference dataloader
(oracle/ingest_oracle.py): the raw rows it stages, the byte masks, `num`, `sample_idx`.  The device half (zero fill of
padded / masked rows) is covered by tests/test_gpu_ingest.py; without a GPU `upload` must refuse to run."""
import numpy as np
import pytest
import torch

import os

import gvd_amd
from gvd_amd import ingest
from oracle import cases, ingest_oracle as IO, ref_harness


@pytest.fixture(scope='module')
def dataset(tmp_path_factory):
    opt = gvd_amd.opts.default_opt(t_attn_size=480)
    root = str(tmp_path_factory.mktemp('feats'))
    fr, sr, recs = IO.write_synthetic_dataset(root, opt, seed=3)
    return opt, fr, sr, recs


@pytest.mark.parametrize('exclude_bgd', [False, True])
def test_staging_matches_reference_dataloader(dataset, exclude_bgd):
    opt, fr, sr, recs = dataset
    want = IO.assemble_batch(recs, fr, sr, opt, exclude_bgd_det=exclude_bgd)
    ing = ingest.InferenceIngest(opt, fr, sr, device=None, max_batch=len(recs), exclude_bgd_det=exclude_bgd, workers=3)
    slot = ing.stage(recs)
    B = len(recs)
    assert slot.B == B
    assert torch.equal(slot.num[:B], want['num'])
    assert torch.equal(slot.sidx[:B], want['sample_idx'])
    assert torch.equal(slot.mask[:B], want['pnt_mask'])           # this batch holds a full-length sample: Rb == R
    for b in range(B):
        n, f = slot.n_pps[b], slot.n_frm[b]
        keep = want['pnt_mask'][b, 1:1 + n] == 0
        # valid, unmasked rows are exactly the reference's; masked / padded rows are the device kernel's job
        assert torch.equal(slot.feat[b, :n][keep], want['ppls_feat'][b, :n][keep])
        assert torch.equal(slot.ppls[b, :n][keep], want['ppls'][b, :n][keep])
        assert torch.equal(slot.segs[b, :f], want['segs_feat'][b, :f])
        assert int(slot.fmask[b, :f].sum()) == 0 and int(slot.fmask[b, f:].sum()) == opt.t_attn_size - f
        assert float(want['segs_feat'][b, f:].abs().sum()) == 0.0


def test_upload_needs_the_gpu(dataset):
    opt, fr, sr, recs = dataset
    ing = ingest.InferenceIngest(opt, fr, sr, device=None, max_batch=2)
    slot = ing.stage(recs[:2])
    with pytest.raises(RuntimeError):
        ing.upload(slot)


def test_sample_idx_rounding_is_half_to_even():
    """dataloader_anet.py:207 uses np.round (banker's rounding): F*t/dur = 2.5 -> 2, 3.5 -> 4."""
    assert np.round(2.5) == 2 and np.round(3.5) == 4


@pytest.mark.skipif(not ref_harness.reference_available(), reason='no /root/reference here')
@pytest.mark.parametrize('exclude_bgd', [False, True])
def test_oracle_equals_the_real_reference_dataloader(dataset, exclude_bgd):
    """THE PIN of oracle/ingest_oracle.py: the reference's real DataLoader.__getitem__ (dataloader_anet.py:175-354, run
    through oracle/ref_dataloader_harness.py) + main.py's trimming vs the restatement, all eleven tensors, bit for bit."""
    from oracle import ref_dataloader_harness as RH
    opt, fr, sr, recs = dataset
    vocab = IO.synthetic_vocab(opt)
    ds = RH.build_reference_dataset(recs, vocab, fr, sr, opt, exclude_bgd_det=exclude_bgd)
    for idx in (list(range(len(recs))), [2], [1, 4]):
        ref = RH.reference_batch(ds, idx, train=True)
        ours = IO.assemble_train_batch([recs[i] for i in idx], vocab, fr, sr, opt, exclude_bgd_det=exclude_bgd)
        for k in cases.INGEST_KEYS:
            assert ref[k].dtype == ours[k].dtype and tuple(ref[k].shape) == tuple(ours[k].shape), k
            assert torch.equal(ref[k], ours[k]), k
        inf = IO.assemble_batch([recs[i] for i in idx], fr, sr, opt, exclude_bgd_det=exclude_bgd)
        refi = RH.reference_batch(ds, idx, train=False)
        for k in ('segs_feat', 'num', 'ppls', 'ppls_feat', 'sample_idx', 'pnt_mask'):
            if k == 'num':      # eval batches carry the box count too (the loader does not know the mode); the model ignores it
                assert torch.equal(refi[k][:, [0, 1, 3, 4, 5, 6]], inf[k][:, [0, 1, 3, 4, 5, 6]])
            else:
                assert torch.equal(refi[k], inf[k]), k


@pytest.mark.parametrize('name', [n for n, s in cases.CASES.items() if s['mode'] == 'ingest'])
def test_oracle_matches_reference_dataloader_golden(name, golden_dir, tmp_path):
    """The travelling pin: tests/golden/ingest_*.npz holds the reference dataloader's outputs on the seeded dataset."""
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    opt, vocab, fr, sr, recs = cases.build_ingest_case(name, str(tmp_path))
    assert len(recs) == int(g['n_records'])
    ours = IO.assemble_train_batch(recs, vocab, fr, sr, opt)
    for k in cases.INGEST_KEYS:
        assert tuple(ours[k].shape) == tuple(g['shape_' + k]), k
        assert cases._bits_checksum(ours[k]) == int(g['fp_' + k]), k
        if k in cases.INGEST_SMALL:
            assert np.array_equal(ours[k].numpy(), g[k]), k


def test_train_ingest_host_half_matches_oracle(dataset):
    """TrainIngest's caption / box staging (product code) vs the pinned oracle."""
    opt, fr, sr, recs = dataset
    vocab = IO.synthetic_vocab(opt)
    want = IO.assemble_train_batch(recs, vocab, fr, sr, opt)
    ing = ingest.TrainIngest(opt, fr, sr, vocab, device=None, max_batch=len(recs), workers=2)
    slot = ing.stage(recs)
    assert torch.equal(slot.num[:len(recs)], want['num'])
    NB = want['gt_boxes'].shape[1]
    for b, (seq, gts, boxes, bmask) in enumerate(slot.train):
        k = boxes.shape[0]
        assert np.array_equal(seq, want['seq'][b].numpy()) and np.array_equal(gts, want['gt_seq'][b].numpy())
        assert np.array_equal(boxes, want['gt_boxes'][b, :k].numpy()) and float(want['gt_boxes'][b, k:].abs().sum()) == 0
        assert np.array_equal(bmask, want['mask_boxes'][b, :, :k].numpy()) and bool((want['mask_boxes'][b, :, k:] == 1).all())
        assert k <= NB
