"""Option surface read by the model (the subset of the reference's argparse flags the hot path uses).

Mirrors the attribute names of /root/reference/opts.py:10-163 that misc/model.py:31-53,55-58,126,137,
145-156,222 and misc/AttModel.py:25-26,63-68,114-131 read, with the reference defaults, so a driver
written against the reference's `opt` namespace can construct this model unchanged.  Data-derived
attributes (vocab_size, itod, wtoi, glove_*, vg_cls) that main.py:599-613 injects at run time are filled
with synthetic stand-ins by `default_opt` and can be overridden by the caller.
"""
import argparse

import torch

# reference defaults (opts.py:38-52, 86, 100) plus the synthetic data-derived sizes
_DEFAULTS = dict(
    rnn_size=1024, num_layers=1, input_encoding_size=512, att_hid_size=512,
    fc_feat_size=3072, att_feat_size=2048, t_attn_size=480, num_sampled_frm=10,
    num_prop_per_frm=100, prop_thresh=0.2, att_model='topdown', att_input_mode='both',
    t_attn_mode='bigru', transfer_mode='cls', region_attn_mode='mix', enable_BUTD=False,
    obj_interact=True, w_att2=0.05, w_grd=0.0, w_cls=0.1, drop_prob_lm=0.5, seq_per_img=1,
    seq_length=20, beam_size=1, test_mode=False, enable_visdom=False, visdom_server='', id='',
    grad_clip=0.1, learning_rate=5e-4, optim='adam', optim_alpha=0.9, optim_beta=0.999,
    weight_decay=0.0, max_epochs=40, learning_rate_decay_start=1, learning_rate_decay_every=3,
    learning_rate_decay_rate=0.8, val_every_epoch=2, inference_only=False, disp_interval=100,
    # data-derived in the reference (dataloader_anet.py:58,126); synthetic here
    vocab_size=5000, detect_size=432,
)


def default_opt(**overrides):
    """Build an `opt` namespace with reference defaults + synthetic data-derived fields.

    `wtoi['UNK']` follows main.py/model.py:53 (`int(opt.wtoi['UNK'])`): a str-valued index.
    """
    d = dict(_DEFAULTS)
    d.update(overrides)
    opt = argparse.Namespace(**d)
    V, D = opt.vocab_size, opt.detect_size
    if not hasattr(opt, 'wtoi'):
        opt.wtoi = {'UNK': str(V - 1)}
    if not hasattr(opt, 'itod'):
        opt.itod = {i: 'cls%d' % i for i in range(1, D + 1)}
    if not hasattr(opt, 'vg_cls'):
        opt.vg_cls = ['vg%d' % i for i in range(1601)]
    g = torch.Generator().manual_seed(1234)
    if not hasattr(opt, 'glove_vg_cls'):
        opt.glove_vg_cls = torch.randn(1601, 300, generator=g)
    if not hasattr(opt, 'glove_clss'):
        opt.glove_clss = torch.randn(D + 1, 300, generator=g)
    return opt
