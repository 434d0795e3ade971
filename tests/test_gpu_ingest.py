"""-m gpu: the whole ingest pipeline (pinned staging -> async H2D of the valid rows -> gvd_zero_masked_rows) delivers
byte-for-byte the tensors the restated reference dataloader + main.py hand to the model, and the model decodes them."""
import os

import pytest
import torch

import gvd_amd
from gvd_amd import att_model, ingest, ops, synth
from oracle import cases, ingest_oracle as IO

pytestmark = pytest.mark.gpu
KEYS = ('segs_feat', 'num', 'ppls', 'ppls_feat', 'sample_idx', 'pnt_mask')


def test_zero_masked_rows_kernel():
    g = torch.Generator().manual_seed(0)
    for B, N, D, off in ((3, 17, 2048, 1), (2, 5, 7, 1), (4, 9, 3072, 0), (1, 1, 5, 0)):
        x = torch.randn(B, N, D, generator=g)
        mask = (torch.rand(B, N + off, generator=g) < 0.4).to(torch.uint8)
        want = x.masked_fill(mask[:, off:off + N].bool().unsqueeze(-1), 0.)
        got = ops.zero_masked_rows(x.cuda().contiguous(), mask.cuda(), mask_off=off).cpu()
        assert torch.equal(got, want)


@pytest.mark.parametrize('Ft', [480, 10])
def test_pipeline_equals_reference_dataloader(tmp_path, Ft):
    opt = gvd_amd.opts.default_opt(t_attn_size=Ft, vocab_size=600)
    fr, sr, recs = IO.write_synthetic_dataset(str(tmp_path), opt, seed=7)
    ing = ingest.InferenceIngest(opt, fr, sr, device=torch.device('cuda', 0), max_batch=4, workers=4)
    seen = 0
    for _ in range(2):                                   # second pass re-uses the staging slots
        for chunk, t in ing.batches(recs, 4):
            want = IO.assemble_batch(chunk, fr, sr, opt)
            torch.cuda.synchronize()
            for k in KEYS:
                assert t[k].dtype == want[k].dtype and tuple(t[k].shape) == tuple(want[k].shape), k
                assert torch.equal(t[k].cpu(), want[k]), k
            seen += len(chunk)
    assert seen == 2 * len(recs)
    # a batch made only of the short sample is trimmed to its proposal count (main.py:339-341)
    short = [r for r in recs if r['proposals'].shape[0] < opt.num_sampled_frm * opt.num_prop_per_frm]
    assert short
    t = ing.upload(ing.stage(short[:1]))
    want = IO.assemble_batch(short[:1], fr, sr, opt)
    assert t['ppls_feat'].shape[1] == short[0]['proposals'].shape[0]
    for k in KEYS:
        assert torch.equal(t[k].cpu(), want[k]), k


def test_ingested_batch_decodes_like_the_oracle_batch(tmp_path):
    opt = gvd_amd.opts.default_opt(t_attn_size=12, vocab_size=600)
    fr, sr, recs = IO.write_synthetic_dataset(str(tmp_path), opt, seed=9, short_props=False)
    sd = synth.init_state_dict(opt, seed=1, profile='trained_like')
    model = att_model.TopDownModel(opt)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    ing = ingest.InferenceIngest(opt, fr, sr, device=torch.device('cuda', 0), max_batch=3)
    chunk, t = next(iter(ing.batches(recs[:3], 3)))
    want = IO.assemble_batch(chunk, fr, sr, opt)
    order = ('segs_feat', 'ppls', 'num', 'ppls_feat', 'sample_idx', 'pnt_mask')
    with torch.no_grad():
        a = model._sample(*[t[k] for k in order])
        b = model._sample(*[want[k].cuda() for k in order])
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])


def test_eval_split_writes_the_reference_result_files(tmp_path):
    """driver.eval_split = the inference loop of main.eval (main.py:313-452) over the ingest pipeline: sentences per
    video in segment order, grounding boxes for words with a detection lemma, the two JSON files."""
    import json
    from gvd_amd import driver
    opt = gvd_amd.opts.default_opt(t_attn_size=12, vocab_size=600)
    opt.id = 'unit'
    fr, sr, recs = IO.write_synthetic_dataset(str(tmp_path / 'feats'), opt, seed=4, short_props=False)
    sd = synth.init_state_dict(opt, seed=2, profile='trained_like')
    model = att_model.TopDownModel(opt)
    model.load_state_dict(sd)
    model = model.cuda()
    itow = {str(i): 'w%d' % i for i in range(opt.vocab_size)}
    wtol = {w: w for w in itow.values()}
    lemma_det = {'w%d' % i: i % 7 + 1 for i in range(1, opt.vocab_size, 3)}
    itod = {i: 'cls%d' % i for i in range(1, 9)}
    ing = ingest.InferenceIngest(opt, fr, sr, device=torch.device('cuda', 0), max_batch=4)
    pred, grd = driver.eval_split(model, ing, recs, 4, itow, opt, wtol=wtol, lemma_det_dict=lemma_det, itod=itod,
                                  out_dir=str(tmp_path / 'out'))
    assert sum(len(v) for v in pred.values()) == len(recs)
    with open(tmp_path / 'out' / 'densecap-validation-unit.json') as f:
        dc = json.load(f)
    assert dc['version'] == 'VERSION 1.0' and set(dc['results']) == set(pred)
    with open(tmp_path / 'out' / 'attn-gen-sent-results-validation-unit.json') as f:
        ag = json.load(f)
    assert ag['eval_mode'] == 'gen'
    # the same sentences as decoding the oracle-assembled batch directly
    want = IO.assemble_batch(recs[:4], fr, sr, opt)
    order = ('segs_feat', 'ppls', 'num', 'ppls_feat', 'sample_idx', 'pnt_mask')
    with torch.no_grad():
        seq = model._sample(*[want[k].cuda() for k in order])[0]
    sents = driver.decode_sequence(itow, seq.cpu())
    got = []
    for r in recs[:4]:
        vid, k = r['seg_id'].split('_segment_')
        got.append((vid, int(k)))
    flat = {}
    for vid, entries in pred.items():
        for j, e in enumerate(entries):
            flat[(vid, j)] = e['sentence']
    assert [flat[g] for g in got] == sents
    for vid in grd:
        for seg, d in grd[vid].items():
            assert len(d['clss']) == len(d['idx_in_sent']) == len(d['bbox_for_all_frames'])
            for boxes in d['bbox_for_all_frames']:
                assert len(boxes) == opt.num_sampled_frm and len(boxes[0]) == 4
    # the pipelined form (file reads, upload, preamble and token loop of neighbouring batches in flight together; batches of
    # 2 -> several of them, more than the in-flight bound): the same sentences and grounding boxes
    ing2 = ingest.InferenceIngest(opt, fr, sr, device=torch.device('cuda', 0), max_batch=2)
    pred2, grd2 = driver.eval_split(model, ing2, recs, 2, itow, opt, wtol=wtol, lemma_det_dict=lemma_det, itod=itod,
                                    pipelined=True, eval_opt={'sample_max': 1, 'beam_size': 1, 'max_in_flight': 2})
    assert dict(pred2) == dict(pred) and dict(grd2) == dict(grd)


def test_bench_files_to_captions_section():
    """bench.py's integrated measurement (synthetic split on disk -> ingest -> pipelined decode -> densecap JSON) runs end to
    end and reports the composed rate next to the two stages alone."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    r = bench.section_files_to_captions(torch.device('cuda', 0), n_seg=24, B=8)
    assert 'error' not in r and 'skipped' not in r, r
    assert r['segments'] == 24 and r['captions_per_s'] > 0 and r['ingest_alone_segments_per_s'] > 0
    assert r['decode_alone_captions_per_s'] > 0


@pytest.mark.parametrize('name', [n for n, s in cases.CASES.items() if s['mode'] == 'ingest'])
def test_train_pipeline_equals_reference_dataloader(name, golden_dir, tmp_path):
    """TrainIngest (pinned staging, async H2D, padding / masking and the frame mask on the GPU) delivers bit for bit the
    eleven tensors the reference's REAL dataloader + main.py produce (tests/golden/ingest_*.npz), and they train."""
    import numpy as np
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    opt, vocab, fr, sr, recs = cases.build_ingest_case(name, str(tmp_path))
    ing = ingest.TrainIngest(opt, fr, sr, vocab, device=torch.device('cuda', 0), max_batch=len(recs), workers=4)
    (chunk, t), = list(ing.batches(recs, len(recs)))
    torch.cuda.synchronize()
    for k in cases.INGEST_KEYS:
        assert tuple(t[k].shape) == tuple(g['shape_' + k]), k
        assert cases._bits_checksum(t[k].cpu()) == int(g['fp_' + k]), k
        if k in cases.INGEST_SMALL:
            assert np.array_equal(t[k].cpu().numpy(), g[k]), k
    # and against the oracle on a sub-batch (different trimming: NB / Rb are batch maxima)
    sub = [recs[2], recs[0]]
    want = IO.assemble_train_batch(sub, vocab, fr, sr, opt)
    got = ing.upload(ing.stage(sub))
    for k in cases.INGEST_KEYS:
        assert torch.equal(got[k].cpu(), want[k]), k
    # the batch drives a training-mode forward (classes of the synthetic captions index the detection vocabulary)
    opt.detect_size = 432
    sd = synth.init_state_dict(opt, seed=2)
    model = att_model.TopDownModel(opt)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    with torch.no_grad():
        losses = model(*[t[k] for k in synth.FORWARD_ORDER], 'MLE')
    assert all(torch.isfinite(l).all() for l in losses[:2])
