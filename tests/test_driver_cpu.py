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


@pytest.mark.skipif(not ref_harness.reference_available(), reason='no /root/reference here')
def test_constructor_equals_reference_constructor_under_the_same_seed():
    """Boundary: `TopDownModel(opt)` draws its parameters in the reference's order and performs the
    Detectron / GloVe knowledge transfer (model.py:173-216) -> a freshly constructed model has the reference's
    state_dict bit for bit under the same torch seed and the same data/detectron_weights pickles."""
    import copy
    opt = gvd_amd.opts.default_opt(vocab_size=97, t_attn_size=6)
    ref, workdir = ref_harness.construct_reference_fresh(copy.deepcopy(opt), seed=5)
    cwd = os.getcwd()
    os.chdir(workdir)
    try:
        torch.manual_seed(5)
        ours = att_model.TopDownModel(copy.deepcopy(opt))
    finally:
        os.chdir(cwd)
    rsd, osd = ref.state_dict(), ours.state_dict()
    assert list(rsd.keys()) == list(osd.keys())
    for k in rsd:
        assert torch.equal(rsd[k], osd[k]), k
    assert torch.equal(ours.matched_cls, ref.matched_cls) and torch.equal(ours.max_sim, ref.max_sim)
    # the transferred tensors really come from the pickles, not from the default init
    assert float(osd['ctx2pool_grd.0.weight'].abs().max()) < 0.1 and float(osd['vis_embed.0.weight'].abs().max()) < 0.1


def test_transfer_mode_none_needs_only_the_fc7_pickles(tmp_path):
    """model.py:173-180,214-215: under transfer_mode='none' the reference reads fc7_w / fc7_b and nothing else - a setup that
    ships only that pair must still initialise `ctx2pool_grd` from it (it used to be skipped without all four pickles)."""
    import pickle
    import numpy as np
    d = tmp_path / 'data' / 'detectron_weights'
    d.mkdir(parents=True)
    rng = np.random.default_rng(2)
    fc7_w = (rng.standard_normal((2048, 2048)) * 0.01).astype(np.float32)
    fc7_b = (rng.standard_normal(2048) * 0.01).astype(np.float32)
    for n, a in (('fc7_w', fc7_w), ('fc7_b', fc7_b)):
        with open(d / (n + '.pkl'), 'wb') as f:
            pickle.dump(a, f)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        m = att_model.TopDownModel(gvd_amd.opts.default_opt(vocab_size=50, transfer_mode='none'))
        # 'cls' still needs the class-score pair: without it the default initialisation stays
        att_model.TopDownModel._warned_no_transfer = True
        c = att_model.TopDownModel(gvd_amd.opts.default_opt(vocab_size=50, transfer_mode='cls'))
    finally:
        os.chdir(cwd)
    assert np.array_equal(m.ctx2pool_grd[0].weight.detach().numpy(), fc7_w)
    assert np.array_equal(m.ctx2pool_grd[0].bias.detach().numpy(), fc7_b)
    assert m.matched_cls is None and not np.array_equal(c.ctx2pool_grd[0].weight.detach().numpy(), fc7_w)


def test_constructor_rejects_dimensions_the_kernels_are_not_built_for():
    for kw in (dict(rnn_size=512), dict(att_hid_size=1024), dict(input_encoding_size=600), dict(seq_length=100),
               # the reference itself raises for these two (profiles/r06/reference_dim_survey.json)
               dict(att_feat_size=4096), dict(fc_feat_size=4096)):
        with pytest.raises(NotImplementedError):
            att_model.TopDownModel(gvd_amd.opts.default_opt(vocab_size=50, **kw))
    # att_hid_size / input_encoding_size BELOW the built widths construct with the reference's parameter shapes (they run through
    # zero-padded operands: goldens *_a256e300) and hand the kernels operands of the built widths
    m = att_model.TopDownModel(gvd_amd.opts.default_opt(vocab_size=50, att_hid_size=256, input_encoding_size=300))
    sd = m.state_dict()
    assert sd['ctx2pool.weight'].shape == (256, 1024) and sd['core.attention.alpha_net.weight'].shape == (1, 256)
    assert sd['embed.0.weight'].shape == (50, 300) and sd['core.att_lstm.weight_ih'].shape == (4096, 1024 + 300)
    with torch.no_grad():
        P = m._decode_params()
    assert P['embed'].shape == (50, 512) and P['att_w_ih'].shape == (4096, 1536) and P['att1_h2att_w'].shape == (512, 1024)
    assert P['att2_alpha_w'].shape == (1, 512) and not P['att2_alpha_w'][:, 256:].any() and not P['embed'][:, 300:].any()
    assert torch.equal(P['att_w_ih'][:, :1324], sd['core.att_lstm.weight_ih']) and not P['att_w_ih'][:, 1324:].any()
    # option values the REFERENCE itself cannot run (profiles/r05/reference_option_survey.json) are rejected ...
    for kw in (dict(att_input_mode='dual_region'), dict(region_attn_mode='add'), dict(region_attn_mode='cat'),
               dict(transfer_mode='glove'), dict(transfer_mode='both'), dict(t_attn_mode='gru')):
        with pytest.raises(NotImplementedError):
            att_model.TopDownModel(gvd_amd.opts.default_opt(vocab_size=50, **kw))


def test_every_option_value_the_reference_runs_constructs_with_the_reference_state_dict_layout():
    """... and every value it CAN run is built: the module's state_dict has exactly the keys / shapes synth.init_state_dict
    draws for that option (= the reference's layout: no alpha_net under 'dp', 4-gate recurrent weights under 'bilstm', no
    vis_classifiers_bias under transfer_mode='none')."""
    from gvd_amd import synth
    for kw in (dict(att_input_mode='featmap'), dict(att_input_mode='region'), dict(region_attn_mode='mix_mul'),
               dict(region_attn_mode='dp'), dict(t_attn_mode='bilstm'), dict(transfer_mode='none')):
        opt = gvd_amd.opts.default_opt(vocab_size=50, **kw)
        m = att_model.TopDownModel(opt)
        want = synth.init_state_dict(opt, seed=1)
        have = m.state_dict()
        assert set(have) == set(want), (kw, set(have) ^ set(want))
        for k in want:
            assert have[k].shape == want[k].shape, (kw, k)
        m.load_state_dict(want)


def test_gt_grounding_results_and_class_accuracy():
    """collect_gt_grounding / class_accuracy vs the literal expressions of main.py:128-153,166-171."""
    g = torch.Generator().manual_seed(3)
    B, L, T, P, V = 3, 6, 2, 5, 40
    opt = argparse.Namespace(num_sampled_frm=T, num_prop_per_frm=P, vocab_size=V, id='t')
    itod = {i: 'cls%d' % i for i in range(1, 10)}
    ppls = torch.rand(B, T * P, 7, generator=g)
    iseq = torch.randint(1, V, (B, 1, L + 1, 4), generator=g)
    iseq[0, 0, 2, 0] = V + 3
    iseq[0, 0, 5, 0] = V + 1
    iseq[2, 0, 1, 0] = V + 7
    att2_ind = torch.randint(0, P, (B, L, T), generator=g)
    grd_ind = torch.randint(0, P, (B, L, T), generator=g)
    a, gr, vocab = driver.collect_gt_grounding(att2_ind, grd_ind, iseq, ppls, ['v1_segment_00', 'v1_segment_01', 'v2_segment_5'],
                                               opt, itod)
    ref_att2 = torch.gather(ppls.view(-1, T, P, 7).permute(0, 2, 1, 3).contiguous(), 1,
                            att2_ind.unsqueeze(-1).expand((B, L, T, 7)))
    assert vocab == {'cls3', 'cls1', 'cls7'}
    assert a['v1']['0']['clss'] == ['cls3', 'cls1'] and a['v1']['0']['idx_in_sent'] == [1, 4]
    assert a['v1']['0']['bbox_for_all_frames'][1] == ref_att2[0, 4, :, :4].tolist()
    assert a['v1']['1'] == {'clss': [], 'idx_in_sent': [], 'bbox_for_all_frames': []}
    assert gr['v2']['5']['clss'] == ['cls7'] and len(gr['v2']['5']['bbox_for_all_frames'][0]) == T
    cls_pred = torch.tensor([[3, 3], [3, 2], [1, 1], [7, 0]])
    acc, n = driver.class_accuracy(cls_pred, vocab)
    assert n == 3 and abs(acc - (0.5 + 1.0 + 0.0) / 3) < 1e-12


def test_run_epochs_schedule_and_checkpoints(tmp_path):
    """Epoch loop of main.py:678-743: LR decay epochs, validation cadence, best-score checkpointing."""
    opt = gvd_amd.opts.default_opt(vocab_size=30, id='ep', max_epochs=9, learning_rate_decay_start=1,
                                   learning_rate_decay_every=3, learning_rate_decay_rate=0.5, val_every_epoch=2,
                                   learning_rate=1.0)
    lin = torch.nn.Linear(2, 2)

    class FakeTrainer:
        model = lin
        optimizer = torch.optim.SGD([{'params': [lin.weight], 'lr': 1.0}, {'params': [lin.bias], 'lr': 0.1}])

        def step(self, args):
            return torch.tensor([1.0, 2.0, 3.0, 4.0])
    scores = {0: 0.3, 2: 0.5, 4: 0.4, 6: 0.6, 8: 0.6}
    seen = []

    def validate(epoch):
        seen.append((epoch, FakeTrainer.optimizer.param_groups[0]['lr'], FakeTrainer.optimizer.param_groups[1]['lr']))
        return {'CIDEr': scores[epoch]}
    infos, hist = driver.run_epochs(FakeTrainer(), opt, lambda e: [None, None], validate, checkpoint_path=str(tmp_path),
                                    log=None)
    # decay at epochs 4 and 7 ((epoch - 1) % 3 == 0 and epoch > 1), both groups scaled
    assert [e for e, _, _ in seen] == [0, 2, 4, 6, 8]
    assert [lr for _, lr, _ in seen] == [1.0, 1.0, 0.5, 0.5, 0.25]
    assert abs(seen[-1][2] - 0.025) < 1e-12 and abs(opt.learning_rate - 0.25) < 1e-12
    assert infos['best_val_score'] == 0.6 and infos['epoch'] == 8
    assert sorted(hist['val_result_history']) == [0, 2, 4, 6, 8]
    for f in ('model.pth', 'model-best.pth', 'infos_ep.pkl', 'infos_ep-best.pkl', 'histories_ep.pkl'):
        assert os.path.isfile(os.path.join(str(tmp_path), f))
    import pickle
    with open(os.path.join(str(tmp_path), 'infos_ep-best.pkl'), 'rb') as f:
        assert pickle.load(f)['epoch'] == 6                    # the tie at epoch 8 does not replace the best (strict >)
