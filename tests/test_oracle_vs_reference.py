"""Direct pin: oracle vs the real reference imported from /root/reference (build container only;
skipped on the GPU box where the reference tree does not exist — tests/golden/ carries the pin there)."""
import pytest
import torch

import gvd_amd
from oracle import edge_cases, gvd_oracle as O, ref_harness

pytestmark = pytest.mark.skipif(not ref_harness.reference_available(), reason='no /root/reference here')


@pytest.fixture(scope='module')
def setup():
    opt = gvd_amd.opts.default_opt(vocab_size=600, t_attn_size=12)
    sd = gvd_amd.synth.init_state_dict(opt, seed=11, profile='trained_like')
    ref = ref_harness.build_reference_model(opt, sd).eval()
    return opt, sd, ref


def test_state_dict_layout_is_the_reference_layout(setup):
    opt, sd, ref = setup
    rsd = ref.state_dict()
    assert list(rsd.keys()) == list(sd.keys())
    for k in sd:
        assert tuple(rsd[k].shape) == tuple(sd[k].shape), k


def test_greedy_bitwise(setup):
    opt, sd, ref = setup
    inp = gvd_amd.synth.make_inputs(opt, 3, seed=5, train=False)
    with torch.no_grad():
        seq, att2, sim = ref(*gvd_amd.synth.as_args(inp), 'sample', {'sample_max': 1, 'beam_size': 1})
        oseq, _, oatt2, osim = O.sample_greedy(sd, opt, inp['segs_feat'], inp['num'], inp['ppls'],
                                               inp['ppls_feat'], inp['sample_idx'], inp['pnt_mask'])
    assert torch.equal(seq, oseq)
    assert torch.equal(att2, oatt2)       # same CPU kernels -> bitwise
    assert torch.equal(sim, osim)


def test_mle_and_grd(setup):
    opt, sd, ref = setup
    inp = gvd_amd.synth.trim_to_batch(gvd_amd.synth.make_inputs(opt, 3, seed=6, train=True))
    args = gvd_amd.synth.as_args(inp)
    with torch.no_grad():
        r = ref(*args, 'MLE')
        o = O.forward_train(sd, opt, *args)
        for a, b in zip(r, o[:4]):
            assert abs(a.item() - b.item()) <= 1e-6
        cp, ai, gi = ref(*args, 'GRD')
        ocp, oai, ogi = O.forward_train(sd, opt, *args, eval_obj_ground=True)
    assert torch.equal(cp, ocp) and torch.equal(ai, oai) and torch.equal(gi, ogi)


def test_reference_beam_is_broken(setup):
    """Documents SURVEY.md §0.4: the reference beam path raises, hence 'parity unpinned' for beam."""
    opt, sd, ref = setup
    inp = gvd_amd.synth.make_inputs(opt, 1, seed=5, train=False)
    with torch.no_grad(), pytest.raises(Exception):
        ref(*gvd_amd.synth.as_args(inp), 'sample', {'sample_max': 1, 'beam_size': 3})


@pytest.mark.parametrize('B,K,seed,end', [(2, 3, 5, None), (3, 5, 6, None), (4, 3, 7, (3.0, -1.5)), (3, 5, 8, (3.0, -2.0))])
def test_beam_reference_with_shim_equals_oracle(B, K, seed, end):
    """Beam pin ("reference-with-shim"): the reference's OWN CaptionModel.beam_search / _sample_beam
    (CaptionModelBU.py:24-185, model.py:627-742) run under oracle/ref_harness.beam_shim (drops the two stray core
    arguments, makes .cuda() a no-op — no reference file edited) gives exactly the ids / attended regions of the
    oracle's restatement `O.sample_beam`; `end` reshapes the END logit so beams finish at different steps."""
    opt = gvd_amd.opts.default_opt(vocab_size=600, t_attn_size=12)
    sd = gvd_amd.synth.init_state_dict(opt, seed=seed, profile='trained_like')
    if end:
        sd['logit.weight'][0] *= end[0]
        sd['logit.bias'][0] += end[1]
    ref = ref_harness.build_reference_model(opt, sd).eval()
    inp = gvd_amd.synth.make_inputs(opt, B, seed=seed, train=False)
    seq, lps, att2 = ref_harness.reference_beam_sample(ref, inp, K)
    with torch.no_grad():
        oseq, olps, oatt, _ = O.sample_beam(sd, opt, inp['segs_feat'], inp['num'], inp['ppls'], inp['ppls_feat'],
                                            inp['sample_idx'], inp['pnt_mask'], beam_size=K)
    assert torch.equal(seq, oseq) and torch.equal(att2, oatt)
    assert float((lps - olps).abs().max()) <= 1e-5
    # the shim is gone afterwards: the unrepaired reference raises again
    with torch.no_grad(), pytest.raises(Exception):
        ref(*gvd_amd.synth.as_args(inp), 'sample', {'sample_max': 1, 'beam_size': K})


@pytest.mark.parametrize('name', sorted(edge_cases.EDGE_CASES))
def test_greedy_edge_shapes_bitwise(name):
    """Pins the oracle on the edge shapes of oracle/edge_cases.py (the GPU test then compares HIP vs oracle)."""
    opt, sd, inp = edge_cases.EDGE_CASES[name]()
    ref = ref_harness.build_reference_model(opt, sd).eval()
    with torch.no_grad():
        seq, att2, sim = ref(*gvd_amd.synth.as_args(inp), 'sample', {'sample_max': 1, 'beam_size': 1})
    oseq, _, oatt2, osim = edge_cases.oracle_greedy(opt, sd, inp)
    assert torch.equal(seq, oseq) and torch.equal(att2, oatt2) and torch.equal(sim, osim)
    if name == 'immediate_end':
        assert int(seq.abs().sum()) == 0


@pytest.mark.parametrize('name', sorted(edge_cases.TRAIN_EDGE_CASES))
def test_mle_edge_shapes(name):
    opt, sd, inp = edge_cases.TRAIN_EDGE_CASES[name]()
    ref = ref_harness.build_reference_model(opt, sd).eval()
    args = gvd_amd.synth.as_args(inp)
    with torch.no_grad():
        r = ref(*args, 'MLE')
        o = O.forward_train(sd, opt, *args)
    for a, b in zip(r, o[:4]):
        assert torch.isfinite(a).all()
        assert abs(a.item() - b.item()) <= 1e-6
