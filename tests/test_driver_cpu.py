"""Driver-side contract (driver.py) against the reference where it is importable: decode_sequence
(misc/utils.py), the grounding-box gather expression of main.py:364-368, and checkpoint interchange with the
reference model (both directions, strict state_dict)."""
import argparse
import os
import sys

import pytest
import torch

import gvd_amd
from gvd_amd import att_model, driver, synth
from oracle import ref_harness


def test_grounding_boxes_expression():
    g = torch.Generator().manual_seed(0)
    B, L, T, P = 3, 20, 10, 100
    att2 = torch.randn(B, L, T * P, generator=g)
    ppls = torch.rand(B, T * P, 7, generator=g)
    ind, boxes = driver.grounding_boxes(att2, ppls, T, P)
    # the literal expression of main.py:364-368
    ref_ind = torch.max(att2.view(B, att2.size(1), T, P), dim=-1)[1]
    ref_box = torch.gather(ppls.view(-1, T, P, 7).permute(0, 2, 1, 3).contiguous(), 1,
                           ref_ind.unsqueeze(-1).expand((B, ref_ind.size(1), T, ppls.size(-1))))
    assert torch.equal(ind, ref_ind) and torch.equal(boxes, ref_box)
    assert torch.equal(ind, att_model.attended_region_indices(att2, T, P))
    b, l, t = 1, 7, 4
    assert torch.equal(boxes[b, l, t], ppls[b, t * P + ind[b, l, t]])


def test_collect_predictions_structure():
    itow = {str(i): 'w%d' % i for i in range(1, 50)}
    seq = torch.tensor([[3, 4, 5, 0, 9], [7, 0, 0, 0, 0]])
    opt = argparse.Namespace(num_sampled_frm=2, num_prop_per_frm=3)
    att2 = torch.randn(2, 5, 6)
    ppls = torch.rand(2, 6, 7)
    wtol = {w: w for w in itow.values()}
    preds, grd = driver.collect_predictions(seq, ['v_a_segment_01', 'v_b_segment_3'], itow,
                                            timestamps={'v_a': {'1': [0.123, 4.567]}, 'v_b': {'3': [1.0, 2.0]}},
                                            att2_weights=att2, ppls=ppls, opt=opt, wtol=wtol,
                                            lemma_det_dict={'w4': 2, 'w7': 5}, itod={2: 'cat', 5: 'dog'})
    assert preds['v_a'][0] == {'sentence': 'w3 w4 w5 ', 'timestamp': [0.12, 4.57]}
    assert preds['v_b'][0]['sentence'] == 'w7 '
    assert grd['v_a']['1']['clss'] == ['cat'] and grd['v_a']['1']['idx_in_sent'] == [1]
    assert len(grd['v_a']['1']['bbox_for_all_frames'][0]) == 2 and grd['v_b']['3']['clss'] == ['dog']


@pytest.mark.skipif(not ref_harness.reference_available(), reason='no /root/reference here')
def test_decode_sequence_matches_reference():
    if ref_harness.REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, ref_harness.REFERENCE_ROOT)
    from misc import utils as ref_utils
    itow = {str(i): 'tok%d' % i for i in range(1, 100)}
    g = torch.Generator().manual_seed(1)
    seq = torch.randint(0, 100, (16, 20), generator=g)
    seq[3] = 0
    seq[5, 0] = 0
    assert driver.decode_sequence(itow, seq) == ref_utils.decode_sequence(itow, None, None, None, None, seq, 100, None)


@pytest.mark.skipif(not ref_harness.reference_available(), reason='no /root/reference here')
def test_checkpoint_interchange_with_reference(tmp_path):
    opt = gvd_amd.opts.default_opt(vocab_size=64, t_attn_size=6, id='unit')
    sd = synth.init_state_dict(opt, seed=3)
    ours = att_model.TopDownModel(opt)
    ours.load_state_dict(sd)
    driver.save_checkpoint(ours, opt, str(tmp_path), infos={'iter': 7, 'epoch': 2, 'best_val_score': 0.5}, best=True,
                           itow={'1': 'a'})
    for f in ('model.pth', 'model-best.pth', 'infos_unit.pkl', 'infos_unit-best.pkl', 'histories_unit.pkl'):
        assert os.path.isfile(os.path.join(str(tmp_path), f))
    # the reference loads our checkpoint exactly like main.py:638 does
    ref = ref_harness.build_reference_model(opt, synth.init_state_dict(opt, seed=9))
    ref.load_state_dict(torch.load(os.path.join(str(tmp_path), 'model.pth')))
    for k, v in ref.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # and we load a checkpoint written the way main.py:711-713 writes it
    torch.save(ref.state_dict(), os.path.join(str(tmp_path), 'model.pth'))
    again = att_model.TopDownModel(opt)
    infos, hist = driver.load_checkpoint(again, str(tmp_path), 'unit')
    assert infos['iter'] == 7 and infos['epoch'] == 2 and infos['vocab'] == {'1': 'a'}
    for k, v in again.state_dict().items():
        assert torch.equal(v, sd[k]), k
