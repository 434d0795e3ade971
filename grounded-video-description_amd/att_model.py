"""Drop-in boundary: `TopDownModel(opt)` with the reference's constructor/forward contract and
`state_dict` layout (misc/AttModel.py:167-171 + misc/model.py:28-234), whose
hot path runs in hand-written HIP kernels (libgvd_hip.so) instead of chains of ATen ops.

    model = TopDownModel(opt).cuda()
    lm, att2, grd, cls = model(segs_feat, seq, gt_seq, num, ppls, gt_boxes, mask_boxes, ppls_feat,
                               frm_mask, sample_idx, pnt_mask, 'MLE')
    seq, att2_weights, sim_mat = model(..., 'sample', {'sample_max': 1, 'beam_size': 1})
    cls_pred, att2_ind, grd_ind = model(..., 'GRD')

What runs where (DESIGN.md §2 has the table): the per-token step (two LSTM cells, both additive
attentions, vocabulary head and token rule), every projection of the per-segment preamble (fc7,
class similarity, pool_embed, the obj_interact encoder incl. its flash-style attention, ctx2pool /
ctx2att, frame embeddings), the bi-GRU frame encoder, the row kernels (class softmax, layer norms),
the target / loss reductions, the whole backward pass, the dropout of the training path and the
optimiser step (optim.ClipAdam) are hand-written HIP kernels of libgvd_hip.so; torch-ROCm library ops
remain for BatchNorm1d, the three few-MB dropout sites (token / visual-word embeddings, h_lang), the
64-row dX products of the token-loop BPTT and autograd's own gradient accumulation.  There is no CPU
path: inputs must live on the GPU.

Supported configuration = the reference's README recipe (att_model='topdown', att_input_mode='both',
region_attn_mode='mix', transfer_mode='cls', t_attn_mode='bigru', seq_per_img=1, enable_BUTD=False) and the other option
values the reference itself can run: att_input_mode 'featmap' / 'region', region_attn_mode 'mix_mul' /
'dp', transfer_mode 'none', t_attn_mode 'bilstm'.
"""
import math
import os
import pickle

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .hip import GvdHipError

MIN_VALUE = -1e8


class _Holder(nn.Module):
    """Plain parameter container (keeps the reference's dotted state_dict names)."""


class _EncLayerNorm(nn.Module):
    """transformer.py:66-77 (unbiased std, eps on std)."""

    def __init__(self, d, eps=1e-6):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(d))
        self.beta = nn.Parameter(torch.zeros(d))
        self.eps = eps

    def forward(self, x):
        mean = x.mean(-1, keepdim=True)
        std = x.std(-1, keepdim=True)
        return self.gamma * (x - mean) / (std + self.eps) + self.beta


def _build_obj_interact(d_model, d_hidden, n_layers, drop_ratio=0.2):
    """Parameter tree of transformer.Transformer as built at model.py:126-135.  The nn.Dropout modules sit where the
    reference has them (ResidualBlock.dropout, transformer.py:84; Attention.dropout, transformer.py:95) so that code
    which walks `model.modules()` to change dropout ratios reaches the same places; they hold no parameters."""
    root = _Holder()
    root.encoder = _Holder()
    layers = []
    for _ in range(n_layers):
        lay = _Holder()
        lay.selfattn = _Holder()
        lay.selfattn.layer = _Holder()
        for n in ('wq', 'wk', 'wv', 'wo'):
            setattr(lay.selfattn.layer, n, nn.Linear(d_model, d_model, bias=False))
        lay.selfattn.layer.attention = _Holder()
        lay.selfattn.layer.attention.dropout = nn.Dropout(drop_ratio)
        lay.selfattn.dropout = nn.Dropout(drop_ratio)
        lay.selfattn.layernorm = _EncLayerNorm(d_model)
        lay.feedforward = _Holder()
        lay.feedforward.layer = _Holder()
        lay.feedforward.layer.linear1 = nn.Linear(d_model, d_hidden)
        lay.feedforward.layer.linear2 = nn.Linear(d_hidden, d_model)
        lay.feedforward.dropout = nn.Dropout(drop_ratio)
        lay.feedforward.layernorm = _EncLayerNorm(d_model)
        layers.append(lay)
    root.encoder.layers = nn.ModuleList(layers)
    return root


class _AttParams(nn.Module):
    def __init__(self, H, A, alpha_net=True):
        super().__init__()
        self.h2att = nn.Linear(H, A)
        if alpha_net:                     # (Attention2 with region_attn_mode='dp' has none: AttModel.py:63-66)
            self.alpha_net = nn.Linear(A, 1)


class _Core(nn.Module):
    """Parameters of TopDownCore (AttModel.py:111-131); the math lives in the HIP library."""

    def __init__(self, opt):
        super().__init__()
        H, E, A = opt.rnn_size, opt.input_encoding_size, opt.att_hid_size
        self.att_lstm = nn.LSTMCell(E + H, H)
        self.lang_lstm = nn.LSTMCell(2 * H, H)
        self.attention = _AttParams(H, A)
        self.attention2 = _AttParams(H, A, alpha_net=opt.region_attn_mode != 'dp')
        self.i2h_2 = nn.Linear(2 * H, H)   # unused in the reference too (AttModel.py:130-131)
        self.h2h_2 = nn.Linear(H, H)


class TopDownModel(nn.Module):
    def __init__(self, opt):
        super().__init__()
        for k, want in (('att_model', ('topdown',)), ('att_input_mode', ('both', 'featmap', 'region')),
                        ('region_attn_mode', ('mix', 'mix_mul', 'dp')),
                        ('transfer_mode', ('cls', 'none')), ('t_attn_mode', ('bigru', 'bilstm')), ('seq_per_img', (1,)),
                        ('enable_BUTD', (False,))):
            if getattr(opt, k) not in want:
                # (profiles/r05/reference_option_survey.json: which other values the REFERENCE itself can run at its README
                # dimensions - region_attn_mode add / cat, att_input_mode dual_region, transfer_mode glove / both raise inside
                # misc/model.py / misc/AttModel.py; every value the reference runs is built here)
                raise NotImplementedError('%s=%r: the HIP path is built for %s (the reference README recipe is the first)'
                                          % (k, getattr(opt, k), ' / '.join(repr(w) for w in want)))
        self.transfer_mode = opt.transfer_mode
        # what the language LSTM is fed (opts.py:58, AttModel.py:140-151): 'both' att + att2 (README), 'featmap' the frame-wise
        # context alone (the region attention still runs: its logits are the grounding output), 'region' the region context
        # alone - no frame-wise encoder / attention at all (model.py:393,406-409).  Same parameters in all three.
        self.att_input_mode = opt.att_input_mode
        # score function of the region attention (opts.py:63, AttModel.py:82-95): 'mix' w . tanh(p + q) (README), 'mix_mul'
        # w . tanh(p * q), 'dp' p . q (no alpha_net in the module, none in the state_dict)
        self.region_attn_mode = opt.region_attn_mode
        self.vocab_size = opt.vocab_size
        self.detect_size = opt.detect_size
        self.rnn_size = H = opt.rnn_size
        self.att_hid_size = opt.att_hid_size
        self.input_encoding_size = opt.input_encoding_size
        self.drop_prob_lm = opt.drop_prob_lm
        self.seq_length = opt.seq_length
        self.seg_info_size = 50
        self.fc_feat_size = opt.fc_feat_size + self.seg_info_size
        self.att_feat_size = opt.att_feat_size
        self.seq_per_img = opt.seq_per_img
        self.num_sampled_frm = opt.num_sampled_frm
        self.num_prop_per_frm = opt.num_prop_per_frm
        self.t_attn_size = opt.t_attn_size
        self.test_mode = opt.test_mode
        self.unk_idx = int(opt.wtoi['UNK'])
        self.vis_encoding_size = 2048
        self.pool_feat_size = self.att_feat_size + 300 + self.detect_size + 1
        self.num_layers = 2
        opt.beta = 1                                  # model.py:72
        self.beta = 1
        D1 = self.detect_size + 1
        p = self.drop_prob_lm

        # parameter layout == reference (model.py:75-161); Sequential indices keep the '.0.' names
        if self.transfer_mode == 'cls':
            self.vis_classifiers_bias = nn.Parameter(torch.zeros(D1))
        else:
            # transfer_mode='none' (model.py:214-215): no class-score transfer, hence NO `vis_classifiers_bias` parameter in the
            # reference (created only at model.py:198; every use is behind hasattr -> bias None, model.py:328-332,472-476) and
            # none in the state_dict here: a zero buffer that is not part of it stands in (x + 0 = x)
            self.register_buffer('vis_classifiers_bias', torch.zeros(D1), persistent=False)
        self.loc_fc = nn.Sequential(nn.Linear(5, 300), nn.ReLU(), nn.Dropout(0.5))
        self.embed = nn.Sequential(nn.Embedding(self.vocab_size, self.input_encoding_size), nn.ReLU(), nn.Dropout(p))
        self.vis_embed = nn.Sequential(nn.Embedding(D1, self.vis_encoding_size), nn.ReLU(), nn.Dropout(p))
        self.fc_embed = nn.Sequential(nn.Linear(self.fc_feat_size, H), nn.ReLU(), nn.Dropout(p))
        self.seg_info_embed = nn.Sequential(nn.Linear(4, self.seg_info_size), nn.ReLU(), nn.Dropout(p))
        self.att_embed = nn.ModuleList([
            nn.Sequential(nn.Linear(2048, H // 2), nn.ReLU(), nn.Dropout(p)),
            nn.Sequential(nn.Linear(1024, H // 2), nn.ReLU(), nn.Dropout(p))])
        self.att_embed_aux = nn.Sequential(nn.BatchNorm1d(H), nn.ReLU())
        self.pool_embed = nn.Sequential(nn.Linear(self.pool_feat_size, H), nn.ReLU(), nn.Dropout(p))
        self.ctx2att = nn.Linear(H, self.att_hid_size)
        self.ctx2pool = nn.Linear(H, self.att_hid_size)
        self.logit = nn.Linear(H, self.vocab_size)
        self.has_obj_interact = bool(opt.obj_interact)
        if self.has_obj_interact:
            self.obj_interact = _build_obj_interact(H, H // 2, 2)
        # frame-wise context encoder (opts.py:60 `--t_attn_mode`, model.py:145-154): 'bigru' (README) or 'bilstm'; the module only
        # holds the parameters (reference names weight_ih_l0, ..._reverse) - the recurrences run in csrc/gru.hip / lstm_seq.hip
        self.t_attn_mode = opt.t_attn_mode
        rnn = nn.LSTM if self.t_attn_mode == 'bilstm' else nn.GRU
        self.context_enc = rnn(H, H // 2, 2, dropout=0.2, bidirectional=True, batch_first=True)
        self.ctx2pool_grd = nn.Sequential(nn.Linear(self.att_feat_size, self.vis_encoding_size), nn.ReLU(),
                                          nn.Dropout(p))
        self._knowledge_transfer(opt)
        self.core = _Core(opt)
        self._validate_dims(opt)

    # widths the decode / BPTT kernels are compiled for (attention.hip: A = 512; decode_persistent.hip, vocab.hip: E = 512)
    A_BUILT, E_BUILT = 512, 512

    def _validate_dims(self, opt):
        """The HIP kernels are specialised at compile time for the reference configuration (attention.hip: A = 512,
        H = 1024; decode_persistent.hip: E = 512; flash kernels: 6 heads of <= 176 columns).  Fail at construction, not
        with GVD_EINVAL in the middle of a forward.  Of the five size options of opts.py:37-48 the reference itself runs
        only three at other values (tools/reference_dim_survey.py -> profiles/r06/reference_dim_survey.json):
        `att_feat_size` != 2048 raises in ITS constructor (the fc7 copy of model.py:177 / the 2048-wide `pool_feats` entering
        `pool_embed`, model.py:312,362) and `fc_feat_size` != 3072 in its forward (`att_embed` is Linear(2048) + Linear(1024),
        model.py:107-110,395).  `att_hid_size` and `input_encoding_size` up to the built widths run here EXACTLY through
        zero-padded operands (_att_axis / _emb_axis below); `rnn_size` is the built 1024 only."""
        bad = []
        if opt.rnn_size != 1024:
            bad.append('rnn_size=%r (built for 1024)' % opt.rnn_size)
        for k, built in (('att_hid_size', self.A_BUILT), ('input_encoding_size', self.E_BUILT)):
            if not 1 <= getattr(opt, k) <= built:
                bad.append('%s=%r (1..%d: narrower widths run zero-padded to the built %d)' % (k, getattr(opt, k), built, built))
        for k, v in (('att_feat_size', 2048), ('fc_feat_size', 3072)):
            if getattr(opt, k) != v:
                bad.append('%s=%r (the reference itself raises for anything but %d: profiles/r06/reference_dim_survey.json)'
                           % (k, getattr(opt, k), v))
        if opt.seq_length > 64 or opt.seq_length < 1:
            bad.append('seq_length=%r (1..64)' % opt.seq_length)
        if opt.num_prop_per_frm * opt.num_sampled_frm < 1:
            bad.append('num_sampled_frm x num_prop_per_frm must be positive')
        if not 0 <= self.unk_idx < self.vocab_size:
            bad.append("wtoi['UNK']=%r outside the vocabulary" % self.unk_idx)
        if bad:
            raise NotImplementedError('TopDownModel: the MI355X kernels of libgvd_hip.so are built for the reference '
                                      'dimensions (opts.py defaults); unsupported: ' + '; '.join(bad))

    # ---- att_hid_size / input_encoding_size below the built widths: zero-padded operands, exact results
    # A projection column a >= att_hid_size has p[r, a] = 0, q[a] = 0 and alpha_net weight 0: it adds 0 * tanh(0) = 0 to every
    # score in every score mode; an embedding column e >= input_encoding_size is relu(0) = 0 and meets zero columns of the
    # att-LSTM's W_ih.  The state_dict keeps the reference's shapes: the padded copies are cached views for inference
    # (_packed) and part of the autograd graph in training (F.pad: the gradient of the pad is the slice).
    def _pad_axis(self, key, t, dim, n):
        if t.shape[dim] == n:
            return t
        pad = [0, 0] * (t.dim() - 1 - dim) + [0, n - t.shape[dim]]
        if torch.is_grad_enabled() and t.requires_grad:
            return F.pad(t, pad)
        return self._packed(('pad', key), (t,), lambda: F.pad(t.detach(), pad).contiguous())

    def _att_proj(self, lin, key):
        """(weight [A_BUILT, H], bias [A_BUILT]) of ctx2pool / ctx2att / h2att."""
        return (self._pad_axis(key + '.w', lin.weight, 0, self.A_BUILT), self._pad_axis(key + '.b', lin.bias, 0, self.A_BUILT))

    def padded_embedding(self, x):
        """relu(embed) rows [.., E] -> [.., E_BUILT] (training path: after the dropout)."""
        return x if x.shape[-1] == self.E_BUILT else F.pad(x, (0, self.E_BUILT - x.shape[-1]))

    def core_params(self):
        """The token loop's parameters at the built widths, reference names -> tensors (inference: cached padded copies;
        under autograd: F.pad of the parameters)."""
        c = self.core
        a1w, a1b = self._att_proj(c.attention.h2att, 'a1')
        a2w, a2b = self._att_proj(c.attention2.h2att, 'a2')
        P = dict(att_w_ih=self._pad_axis('att_w_ih', c.att_lstm.weight_ih, 1, self.rnn_size + self.E_BUILT),
                 a1_w=a1w, a1_b=a1b, a1_aw=self._pad_axis('a1_aw', c.attention.alpha_net.weight, 1, self.A_BUILT),
                 a2_w=a2w, a2_b=a2b)
        if hasattr(c.attention2, 'alpha_net'):            # (none under region_attn_mode='dp', AttModel.py:63-66)
            P['a2_aw'] = self._pad_axis('a2_aw', c.attention2.alpha_net.weight, 1, self.A_BUILT)
        return P

    def _knowledge_transfer(self, opt):
        """model.py:173-216 (`transfer_mode='cls'`): the Detectron fc7 layer initialises `ctx2pool_grd`, and every
        object class takes the Detectron `cls_score` row / bias of its nearest Visual-Genome class (cosine similarity
        of GloVe vectors) as its visual word `vis_embed` / `vis_classifiers_bias`.  Like the reference the pickles
        are read relative to the CWD (`data/detectron_weights/{fc7_w,fc7_b,cls_score_w,cls_score_b}.pkl`); unlike the
        reference (which raises FileNotFoundError) a missing directory leaves the default initialisation in place —
        checkpoints overwrite these tensors anyway — and says so once."""
        d = os.path.join('data', 'detectron_weights')
        # transfer_mode='none' reads only the fc7 pair (model.py:173-180,214-215); the class-score pair belongs to 'cls'
        names = ('fc7_w', 'fc7_b') + (('cls_score_w', 'cls_score_b') if self.transfer_mode == 'cls' else ())
        if not all(os.path.exists(os.path.join(d, n + '.pkl')) for n in names):
            if not getattr(TopDownModel, '_warned_no_transfer', False):
                print('TopDownModel: %s/*.pkl not found - no Detectron knowledge transfer (model.py:173-216); '
                      'load a checkpoint or provide the pickles' % d)
                TopDownModel._warned_no_transfer = True
            self.matched_cls = self.max_sim = None
            return
        load = {}
        for n in names:
            with open(os.path.join(d, n + '.pkl'), 'rb') as f:
                load[n] = torch.from_numpy(pickle.load(f))
        with torch.no_grad():
            self.ctx2pool_grd[0].weight[:self.att_feat_size].copy_(load['fc7_w'])
            self.ctx2pool_grd[0].bias[:self.att_feat_size].copy_(load['fc7_b'])
            if self.transfer_mode == 'none':               # model.py:214-215: only the fc7 layer is transferred
                self.matched_cls = self.max_sim = None
                return
            assert len(opt.itod) + 1 == opt.glove_clss.size(0)        # index 0 is background (model.py:189)
            assert len(opt.vg_cls) == opt.glove_vg_cls.size(0)
            vg = opt.glove_vg_cls / torch.norm(opt.glove_vg_cls, dim=1).unsqueeze(1)
            cl = opt.glove_clss / torch.norm(opt.glove_clss, dim=1).unsqueeze(1)
            max_sim, matched = torch.max(torch.matmul(vg, cl.transpose(1, 0)), dim=0)
            self.max_sim, self.matched_cls = max_sim, matched
            w, b = load['cls_score_w'], load['cls_score_b']
            vis = opt.glove_clss.new_zeros(self.detect_size + 1, w.size(1))
            bias = opt.glove_clss.new_zeros(self.detect_size + 1)
            vis[0], bias[0] = w[0], b[0]                                  # background
            vis[1:] = w[matched[1:]]
            bias[1:] = b[matched[1:]]
            self.vis_embed[0].weight.copy_(vis)
            self.vis_classifiers_bias.copy_(bias)

    # ------------------------------------------------------------------ API (model.py:227-234)
    def forward(self, segs_feat, seq, gt_seq, num, ppls, gt_boxes, mask_boxes, ppls_feat, frm_mask,
                sample_idx, pnt_mask, opt, eval_opt={}):
        if not ppls_feat.is_cuda:
            raise GvdHipError('TopDownModel runs on the MI355X HIP path only; move the model and inputs to the GPU')
        if opt == 'MLE':
            return self._forward_train(segs_feat, seq, gt_seq, ppls, gt_boxes, mask_boxes, num, ppls_feat,
                                       frm_mask, sample_idx, pnt_mask, False)
        if opt == 'GRD':
            return self._forward_train(segs_feat, seq, gt_seq, ppls, gt_boxes, mask_boxes, num, ppls_feat,
                                       frm_mask, sample_idx, pnt_mask, True)
        if opt == 'sample':
            seq, lps, att2, sim = self._sample_checked(segs_feat, ppls, num, ppls_feat, sample_idx, pnt_mask, eval_opt)
            return seq, att2, sim
        raise ValueError(opt)

    # ------------------------------------------------------------------ helpers
    def _flags(self):
        """Device status words of the persistent cooperative kernels launched since the last check."""
        if not hasattr(self, '_kernel_flags'):
            self._kernel_flags = []
        return self._kernel_flags

    def kernel_status_counts(self):
        """Device int64 vector [persistent-kernel barrier timeouts, loader-contract violations] of all launches since the
        last call (no host sync; the lists are cleared), or None when nothing was launched."""
        flags, self._kernel_flags = self._flags(), []
        contract, self._contract_flags = self.__dict__.get('_contract_flags', []), []
        if not flags and not contract:
            return None
        dev = (flags + contract)[0].device
        return torch.stack([torch.stack([f.reshape(-1).ne(0).sum() for f in fl]).sum() if fl else
                            torch.zeros((), dtype=torch.int64, device=dev) for fl in (flags, contract)])

    def _status_host(self):
        """(barrier timeouts, contract violations) of all launches since the last check as host integers, or None when
        nothing was launched: every status word of the call in ONE concatenation and ONE device->host copy (the device-side
        counting of kernel_status_counts is ~12 small launches: 1.5 % of a batch_size = 4 call)."""
        flags, self._kernel_flags = self._flags(), []
        contract, self._contract_flags = self.__dict__.get('_contract_flags', []), []
        if not flags and not contract:
            return None
        words = [f.reshape(-1) for f in flags + contract]
        n_bad = sum(w.numel() for w in words[:len(flags)])
        host = (words[0] if len(words) == 1 else torch.cat(words)).tolist()          # the call's one device->host read
        return sum(1 for v in host[:n_bad] if v), sum(1 for v in host[n_bad:] if v)

    def _sample_checked(self, segs_feat, ppls, num, ppls_feat, sample_idx, pnt_mask, eval_opt):
        """_sample + the status check of its launches (one device->host read per call), with the two conditions a
        reference user never sees COMPUTED instead of raised:
          * masked proposals (pnt_mask = 1) with NON-zero features / boxes - inputs the reference accepts (model.py:311-391
            computes every row) but for which the compacted preamble's premise (all masked rows of a segment are the same
            zero row, dataloader_anet.py:343-344) does not hold: the batch is decoded again through the dense preamble;
          * a grid-barrier timeout of a persistent kernel (workgroups not co-resident, e.g. a shared GPU): the persistent
            kernels are switched off for the process (ops.disable_persistent_kernels) and the batch is decoded again on the
            kernel-per-op decoder / the cooperative GRU launch.
        Only a failure of the retry raises."""
        if self.__dict__.get('_dense_preamble_sticky') and not eval_opt.get('dense_preamble'):
            # an earlier call met masked proposals that are not zero rows: this data source breaks the loader contract, so the
            # compacted preamble would be computed and thrown away on every call (compact + dense = ~2x) - go dense directly
            eval_opt = dict(eval_opt, dense_preamble=True)
        out = self._sample(segs_feat, ppls, num, ppls_feat, sample_idx, pnt_mask, eval_opt)
        st = self._status_host()                    # one device->host read per call
        if st is None or st == (0, 0):
            return out
        bad, contract = st
        if bad:
            ops.disable_persistent_kernels(bad)
        if contract and not self.__dict__.get('_dense_preamble_sticky'):
            self._dense_preamble_sticky = True
            import warnings
            warnings.warn("TopDownModel: masked proposals (pnt_mask = 1) with non-zero features / boxes - inputs the reference "
                          "accepts but that break the loader's zero-row contract (dataloader_anet.py:343-344) the compacted "
                          "preamble relies on; this and every later 'sample' call of this model run on the dense preamble",
                          RuntimeWarning, stacklevel=3)
        opt2 = dict(eval_opt, dense_preamble=True) if contract else eval_opt
        out = self._sample(segs_feat, ppls, num, ppls_feat, sample_idx, pnt_mask, opt2)
        bad2, contract2 = self._status_host() or (0, 0)
        self.raise_for_status(bad2, contract2 if contract else 0)
        return out

    @staticmethod
    def raise_for_status(bad, contract):
        if contract:
            raise GvdHipError('%d batch(es) had masked proposals (pnt_mask = 1) whose fc6 features / boxes are not zero: the '
                              "compacted preamble's results are invalid for them.  forward(..., 'sample') recomputes such "
                              "batches on the dense preamble by itself; direct callers of _sample pass "
                              "{'dense_preamble': True} or set GVD_COMPACT=0" % contract)
        if bad:
            raise GvdHipError('%d persistent-kernel launch(es) hit a grid-barrier timeout (workgroups not co-resident, '
                              'e.g. a shared GPU): results are invalid.  forward() / Trainer.step() recompute such calls without the '
                              'persistent kernels by themselves, sample_pipelined() does when its batches are a list (a lazy '
                              'producer cannot be replayed: driver.eval_split then falls back to its batch-by-batch loop); '
                              'direct callers of _sample call ops.disable_persistent_kernels() and run again' % bad)

    def check_kernel_status(self):
        """The persistent kernels (greedy decoder B <= 4, bi-GRU) bound their grid-barrier spins and latch a flag
        instead of hanging the GPU when workgroups are not co-resident; their outputs are then garbage.  One deferred
        device->host read for all launches since the last call; raises GvdHipError if any flag is set.  Called by
        `forward(..., 'sample')` and `sample_pipelined`; callers of the private `_sample` (bench, tests) call it
        themselves after their timed region."""
        counts = self.kernel_status_counts()
        if counts is not None:
            bad, contract = counts.tolist()      # one device->host read
            self.raise_for_status(bad, contract)

    def _lin(self, x, lin, act=0, p_drop=0.0):
        """nn.Linear (+ReLU (+training-mode dropout of probability p_drop)) on the fp32 MFMA GEMM."""
        return ops.linear(x, lin.weight, lin.bias, act, p_drop)

    def _fused_drop_p(self, p=None):
        """Drop probability to hand to a fused Linear + ReLU + dropout site (ops.linear(..., p_drop)): the module's
        probability in training mode, 0 in eval mode."""
        if not self.training:
            return 0.0
        return float(self.drop_prob_lm if p is None else p)

    def _lin_k32(self, x, lin, act=0, x_padded=False, p_drop=0.0):
        """nn.Linear (+ReLU) whose input width is not a multiple of the GEMM's 32-deep k tile (fc_embed: K = 3122,
        loc_fc: K = 5) on the fp32 MFMA GEMM: input and weight zero-padded along K (the padded weight is cached at
        inference; under autograd the pad is part of the graph).  x_padded: x already carries the zero columns (the fused
        feature kernels of the inference preamble write them)."""
        pad = (-lin.in_features) % 32
        if pad == 0:
            return self._lin(x, lin, act, p_drop)
        xp = x if x_padded else F.pad(x, (0, pad))
        assert xp.shape[-1] == lin.in_features + pad
        if torch.is_grad_enabled():
            w = F.pad(lin.weight, (0, pad))
        else:
            w = self._packed(('k32', id(lin)), (lin.weight,), lambda: F.pad(lin.weight, (0, pad)).contiguous())
        return ops.linear(xp, w, lin.bias, act, p_drop)

    def _drop(self, x, p=None):
        """F.dropout at the reference's stand-alone dropout sites (token / visual-word embeddings model.py:79-82,428,470;
        h_lang AttModel.py:161; seg_info model.py:308): the Philox row kernel forward and backward (ops.dropout) - the ATen op only
        for tensors it does not take (CPU tensors of the control-flow tests, element counts that are not a multiple of 4)."""
        p = self.drop_prob_lm if p is None else p
        if x.is_cuda and x.dtype == torch.float32 and x.numel() % 4 == 0 and x.numel() > 0:
            return ops.dropout(x, p, self.training)
        return F.dropout(x, p, self.training)

    def _packed(self, key, params, build):
        """Derived (re-laid-out) copies of parameters for the fused inference kernels, rebuilt whenever a source
        parameter changed (in-place update, load_state_dict, .to(device))."""
        cache = self.__dict__.setdefault('_pack_cache', {})
        sig = tuple((t.data_ptr(), t._version, t.device) for t in params)
        # nn.DataParallel replicas are shallow copies that share this dict: one entry per device, so that the replicas do
        # not evict each other's packed weights on every call
        key = (key, params[0].device)
        hit = cache.get(key)
        if hit is None or hit[0] != sig:
            with torch.no_grad():
                hit = (sig, build())
            cache[key] = hit
        return hit[1]

    @staticmethod
    def _add_ln(x, y, ln, p_drop=0.0):
        """ResidualBlock tail (transformer.py:79-88): dropout(p_drop, already resolved for the mode) on the branch y, add +
        the custom LayerNorm: one fused row kernel forward and one backward for the d_model the kernels are built for; the
        module's elementwise form otherwise."""
        if x.shape[-1] == 1024 and x.is_cuda:
            return ops.add_layernorm(x, y, ln.gamma, ln.beta, ln.eps, p_drop)
        return ln(x + (F.dropout(y, p_drop, True) if p_drop > 0 else y))

    def _obj_interact_train(self, x, scale, key_bias=None):
        """Training path of the region encoder (transformer.py:39-117): every layer is ONE autograd function
        (ops._EncLayerFn: packed q | k | v projection with every head in its own 176-column slot, flash-style attention core
        with the backward maps only inside the backward, fused residual LayerNorms, feed-forward; hand-scheduled backward
        with the residual gradients folded into the dX products) over the region rows of the batch packed back to back -
        no pad rows when R % 4 == 0 and B R % 32 == 0 (the README shapes), else the region axis zero-padded to a multiple of
        32 for the whole stack (pad rows never reach the loss: their gradients are exactly zero, and they are masked out of
        every softmax)."""
        B, R, d = x.shape
        Rs = ops.enc_layer_rows(B, R)
        xp = (x if Rs == R else F.pad(x, (0, 0, 0, Rs - R))).reshape(B * Rs, d).contiguous()
        for lay in self.obj_interact.encoder.layers:
            xp = ops.enc_layer(xp, lay, B, R, scale, self.training, key_bias)
        xp = xp.view(B, Rs, d)
        return xp if Rs == R else xp[:, :R]

    def _obj_interact_fused(self, x, ci=None):
        """Inference path of the encoder on the HIP kernels only (transformer.py:107-190): per layer ONE projection GEMM
        for q|k|v against row-permuted weights that drop every head into its own zero-padded 176-column slot (16-byte
        aligned heads), the padded-head flash attention kernel, the output projection over the padded layout (zero
        weight columns on the pads), residual + LayerNorm row kernel, and the feed-forward pair on the MFMA GEMM with
        bias/ReLU fused.  With `ci` (ops.CompactIndex) x is the compacted row set [cap, d]: every GEMM / row kernel works
        on the live rows only and the attention runs ragged with the weighted representative key (csrc/compact.hip)."""
        d = x.shape[-1]
        HP, nh = ops.HEAD_PAD, 6
        m = ci.m_dev if ci is not None else None
        ragged = (ci.B, ci.R + 1, ci.off, ci.rep_w) if ci is not None else None
        sizes = [t.shape[-1] for t in x.reshape(-1, d)[:1].chunk(nh, -1)]
        starts = [sum(sizes[:i]) for i in range(nh)]
        for lay in self.obj_interact.encoder.layers:
            sa = lay.selfattn.layer

            def build_qkv(sa=sa):
                w = torch.zeros(3 * nh * HP, d, device=x.device, dtype=torch.float32)
                for j, lin in enumerate((sa.wq, sa.wk, sa.wv)):
                    for h in range(nh):
                        w[(j * nh + h) * HP:(j * nh + h) * HP + sizes[h]] = lin.weight[starts[h]:starts[h] + sizes[h]]
                return w

            def build_wo(sa=sa):
                w = torch.zeros(d, nh * HP, device=x.device, dtype=torch.float32)
                for h in range(nh):
                    w[:, h * HP:h * HP + sizes[h]] = sa.wo.weight[:, starts[h]:starts[h] + sizes[h]]
                return w
            w_qkv = self._packed(('qkv', id(sa)), (sa.wq.weight, sa.wk.weight, sa.wv.weight), build_qkv)
            w_o = self._packed(('wo', id(sa)), (sa.wo.weight,), build_wo)
            qkv = ops.gemm_nt(x, w_qkv, m_dev=m)                               # [B,R,3*6*176]
            o = ops.flash_attn_padded(qkv, nh, 1.0 / math.sqrt(d), ragged=ragged)   # [B,R,6*176]
            att = ops.gemm_nt(o, w_o, m_dev=m)
            ln = lay.selfattn.layernorm
            x = ops.add_layernorm_unbiased(x.contiguous(), att, ln.gamma.detach(), ln.beta.detach(), ln.eps, rows_dev=m)
            ff = lay.feedforward.layer
            y = ops.gemm_nt(ops.gemm_nt(x, ff.linear1.weight.detach(), ff.linear1.bias.detach(), 1, m_dev=m),
                            ff.linear2.weight.detach(), ff.linear2.bias.detach(), m_dev=m)
            ln = lay.feedforward.layernorm
            x = ops.add_layernorm_unbiased(x, y, ln.gamma.detach(), ln.beta.detach(), ln.eps, rows_dev=m)
        return x

    def _fused_encoder_ok(self, d):
        scale = math.sqrt(d)
        return (self.has_obj_interact and scale == 2.0 ** round(math.log2(scale)) and d % 32 == 0
                and -(-d // 6) <= ops.HEAD_PAD)

    def _vis_words_padded(self):
        """relu(vis_embed.weight) and vis_classifiers_bias zero-padded along the class axis to a multiple of 32 (inference;
        cached until a source parameter changes)."""
        w, b = self.vis_embed[0].weight, self.vis_classifiers_bias

        def build(w=w, b=b):
            pad = (-w.shape[0]) % 32
            return (F.pad(F.relu(w), (0, 0, 0, pad)).contiguous(), F.pad(b, (0, pad)).contiguous())
        return self._packed(('vis_words',), (w, b), build)

    def _pool_weight_padded(self, K):
        pw = self.pool_embed[0].weight

        def build_pool(pw=pw, K=K):
            w = torch.zeros(pw.shape[0], K, device=pw.device, dtype=torch.float32)
            w[:, :pw.shape[1]] = pw
            return w
        return self._packed(('pool_embed', K), (pw,), build_pool)

    def _regions_compact(self, ppls, ppls_feat, pm):
        """The region half of the inference preamble (model.py:311-391) on the COMPACTED row set: masked proposals are
        zero rows by the loader contract (dataloader_anet.py:343-344), so per segment only its valid rows plus ONE
        representative masked row are computed (csrc/compact.hip); the dense [B,R,.] tensors the token loop streams
        are restored by a row gather only for the consumers that need them (`_dense_regions`): the greedy token loop reads
        the compacted rows in place through the row map.  -> ci, pool [cap,H], p_pool [cap,A], sim_mat [B,D1,R]."""
        ci = ops.CompactIndex(pm)
        flag = torch.zeros(1, dtype=torch.int32, device=pm.device)
        ops.check_masked_rows_zero(ppls_feat.contiguous(), pm, flag)
        ops.check_masked_rows_zero(ppls.contiguous(), pm, flag)
        self.__dict__.setdefault('_contract_flags', []).append(flag)
        m = ci.m_dev
        fc7 = self.ctx2pool_grd[0]
        if ppls_feat.numel() * 4 < (1 << 32):
            # fc7 reads its rows of the dense fc6 tensor through the compaction map (row gather fused into the GEMM)
            g_pool = ops.gemm_nt(ppls_feat.contiguous(), fc7.weight.detach(), fc7.bias.detach(), 1, m_dev=m,
                                 a_row_map=ci.src_row)                                                      # [cap,2048]
        else:
            g_pool = ops.gemm_nt(ci.gather(ppls_feat), fc7.weight.detach(), fc7.bias.detach(), 1, m_dev=m)
        # location features of the compacted rows in one launch, already padded to the GEMM's 32-deep k tile (model.py:357-360)
        loc_in = ops.loc_features(ppls.contiguous(), ci.src_row, m, ci.cap, self.num_sampled_frm, ldo=32)
        loc = self._lin_k32(loc_in, self.loc_fc[0], act=1, x_padded=True)
        # class logits: the D1 = 433 visual words zero-padded to 448 rows so that the output rows are 16-byte aligned (the
        # GEMM's vectorised LDS epilogue instead of column-strided scalar stores); the row kernel reads D1 of them
        vis_word, vis_bias = self._vis_words_padded()
        logits = ops.gemm_nt(g_pool, vis_word, vis_bias, m_dev=m)                                         # [cap,448]
        pool_in, sim_c = ops.region_feature_rows_compact(g_pool, loc, logits, ci.cmask, m, pad_to=32,
                                                         n_cls=self.detect_size + 1)
        pool = ops.gemm_nt(pool_in, self._pool_weight_padded(pool_in.shape[-1]), self.pool_embed[0].bias.detach(), 1,
                           m_dev=m)
        pool = self._obj_interact_fused(pool, ci=ci)
        w_cp, b_cp = self._att_proj(self.ctx2pool, 'ctx2pool')
        p_pool = ops.gemm_nt(pool, w_cp.detach(), b_cp.detach(), m_dev=m)
        return ci, pool, p_pool, ci.expand(sim_c).transpose(1, 2)

    def _obj_interact(self, x, key_bias=None):
        """transformer.py:135-190,244-254 as built at model.py:126-135 (6 uneven heads, scale sqrt(d_model), no padding
        mask, custom LayerNorm): the flash-style kernels without autograd, the fused MFMA training layers with it (any region
        count up to 4096 padded rows per sample).  Beyond that - and on CPU tensors (the control-flow tests) - the elementwise
        formulation below: projections / feed-forward on the MFMA GEMM, the attention maps through torch (counted by
        ops.library_fallback; an error under GVD_STRICT)."""
        d = x.shape[-1]
        scale = math.sqrt(d)
        if not torch.is_grad_enabled() and not self.training and self._fused_encoder_ok(d):
            return self._obj_interact_fused(x)
        if ((torch.is_grad_enabled() or self.training) and x.is_cuda and scale == 2.0 ** round(math.log2(scale))
                and ops.enc_layer_ok(x.shape[1], d)):
            return self._obj_interact_train(x, scale, key_bias)
        if x.is_cuda:
            ops.library_fallback('encoder attention maps', 'R = %d, d_model = %d' % (x.shape[1], d))
        for lay in self.obj_interact.encoder.layers:
            sa = lay.selfattn.layer
            q, k, v = self._lin(x, sa.wq), self._lin(x, sa.wk), self._lin(x, sa.wv)
            heads = []
            for qh, kh, vh in zip(q.chunk(6, -1), k.chunk(6, -1), v.chunk(6, -1)):
                dots = torch.matmul(qh, kh.transpose(1, 2)) / scale           # transformer.py:92,104
                if key_bias is not None:          # (compacted training layout)
                    dots = dots + key_bias.unsqueeze(1)
                w = F.softmax(dots, dim=-1)
                heads.append(torch.matmul(F.dropout(w, sa.attention.dropout.p, self.training), vh))
            att = self._lin(torch.cat(heads, -1), sa.wo)
            ff = lay.feedforward.layer
            # ResidualBlock (transformer.py:79-88): dropout on the branch, then add + LayerNorm as one fused row kernel
            x = self._add_ln(x, att, lay.selfattn.layernorm, lay.selfattn.dropout.p if self.training else 0.0)
            y = self._lin(self._lin(x, ff.linear1, act=1), ff.linear2)
            x = self._add_ln(x, y, lay.feedforward.layernorm, lay.feedforward.dropout.p if self.training else 0.0)
        return x

    def _preamble(self, segs_feat, num, ppls, ppls_feat, sample_idx, pnt_mask, allow_compact=False, enc_key_bias=None):
        """Per-segment work shared by the three drivers (model.py:302-409 / 504-568 / 634-698).
        enc_key_bias: per-sample key weights of the encoder's self-attention, given when the caller handed in the compacted
        training layout (train_compact.py; the dense training path only)."""
        B, Ft = segs_feat.shape[0], segs_feat.shape[1]
        R = ppls.shape[1]
        D1 = self.detect_size + 1
        pm = pnt_mask if pnt_mask.dtype == torch.uint8 else pnt_mask.to(torch.uint8)
        pm = pm.contiguous()
        # fc feature (model.py:306-308)
        fused_side = (not self.training and not torch.is_grad_enabled() and num.dtype == torch.int64
                      and segs_feat.shape[-1] <= 4096 and self.seg_info_size <= 64)
        if fused_side:
            # inference: mean over frames, seg_info_embed + ReLU, both layer norms, the concat and the K pad of fc_embed as one
            # launch instead of ~10 (they matter at batch_size = 4: csrc/compact.hip)
            fc = ops.fc_feature(segs_feat.contiguous(), num.contiguous(), self.seg_info_embed[0].weight.detach(),
                                self.seg_info_embed[0].bias.detach(), pad_to=32)
        else:
            fc = segs_feat.mean(dim=1)
            # seg_info_embed (model.py:308): Linear(4, 50) + ReLU + dropout on the MFMA GEMM like every other Linear of the path
            # (K = 4 zero-padded to the 32-deep k tile) - a [B,4] x [4,50] library product otherwise
            if segs_feat.is_cuda:
                seg_info = self._drop(self._lin_k32(num[:, 3:7].float().contiguous(), self.seg_info_embed[0], act=1))
            else:
                seg_info = self._drop(F.relu(self.seg_info_embed[0](num[:, 3:7].float())))
            fc = torch.cat([F.layer_norm(fc, [fc.shape[-1]]), F.layer_norm(seg_info, [self.seg_info_size])], dim=-1)
        compact = (allow_compact and not torch.is_grad_enabled() and not self.training
                   and os.environ.get('GVD_COMPACT', '1') == '1'
                   and self._fused_encoder_ok(self.rnn_size) and B * (R + 1) < (1 << 24))
        if compact:
            ci, pool_c, p_pool_c, sim_mat = self._regions_compact(ppls, ppls_feat, pm)
            pre = self._preamble_finish(segs_feat, sample_idx, fc, pm, None, None, sim_mat, None)
            pre.update(ci=ci, pool_c=pool_c, p_pool_c=p_pool_c)
            return pre
        # fc7 over the raw fc6 region features: MFMA GEMM + fused bias/ReLU (model.py:311-313)
        g_pool = self._lin(ppls_feat, self.ctx2pool_grd[0], act=1, p_drop=self._fused_drop_p())
        vis_word = self._drop(F.relu(self.vis_embed[0].weight))
        loc_in = torch.cat([ppls[:, :, :4] / 720., (ppls[:, :, 4] * 1. / self.num_sampled_frm).unsqueeze(-1)], dim=2)
        loc = self._lin_k32(loc_in, self.loc_fc[0], act=1, p_drop=self._fused_drop_p(self.loc_fc[2].p))
        if not torch.is_grad_enabled():
            # inference: class-last similarity logits from ONE plain MFMA GEMM (the visual words are shared by the
            # batch), then mask + class softmax + the three layer norms + concat as one HIP row kernel
            # (model.py:321-340,357-364)
            if self.training:      # (train mode under no_grad: the dropout on the visual words is live, model.py:329)
                cpad = (-D1) % 32
                vw_pad, vb_pad = F.pad(vis_word, (0, 0, 0, cpad)), F.pad(self.vis_classifiers_bias, (0, cpad))
            else:
                vw_pad, vb_pad = self._vis_words_padded()
            logits_t = ops.gemm_nt(g_pool, vw_pad, vb_pad)                                        # [B,R,448] (D1 = 433 used)
            pool, sim_t = ops.region_feature_rows(g_pool, loc.contiguous(), logits_t, pm, pad_to=32, n_cls=D1)
            sim_mat = sim_t.transpose(1, 2)          # [B,D1,R] view (the reference returns this layout)
            # pool_embed (model.py:384, K = 2781) on the MFMA GEMM: the row kernel wrote the concat zero-padded to
            # K = 2784 (16-byte aligned rows, 32-multiple K); the weight gets matching zero columns once
            pool = self._drop(ops.gemm_nt(pool, self._pool_weight_padded(pool.shape[-1]),
                                          self.pool_embed[0].bias.detach(), 1))
        elif g_pool.shape[-1] == 2048 and D1 <= 512 and loc.shape[-1] <= 512:
            # training: the same class-last similarity GEMM + fused row kernel as inference, with a fused backward row
            # kernel (three layer-norm backwards + class-softmax backward, ops._RegionRowsFn).  The class axis is
            # zero-padded to a 32-multiple so that dX / dW of the similarity GEMM run on the K-strided MFMA kernel; the
            # pad classes never enter the softmax (n_cls = D1) and receive zero gradient.
            cpad = (-D1) % 32
            logits_t = ops.linear(g_pool, F.pad(vis_word, (0, 0, 0, cpad)), F.pad(self.vis_classifiers_bias, (0, cpad)))
            pool, sim_t = ops.region_feature_rows_train(g_pool, loc, logits_t, pm, D1, pad_to=32)
            sim_mat = sim_t.transpose(1, 2)          # [B,D1,R] view of the class-last tensor
            pe = self.pool_embed[0]
            pool = ops.linear(pool, F.pad(pe.weight, (0, pool.shape[-1] - pe.weight.shape[1])), pe.bias, 1,
                              self._fused_drop_p())
        else:
            # (class counts beyond what the fused row kernels hold in registers, detect_size > 511) region-class similarity:
            # batched grounder GEMM with fused bias + proposal mask (model.py:321-340), torch row ops
            sim_logits = ops.grounder(vis_word, g_pool, pm[:, 1:], mbias=self.vis_classifiers_bias, xt_shared=True)
            sim_mat = F.softmax(sim_logits, dim=1)
            # location / class-distribution features (model.py:357-364)
            label = sim_mat.permute(0, 2, 1)
            kpad = (-(g_pool.shape[-1] + 300 + D1)) % 32          # zero columns: 32-multiple K for the MFMA GEMM
            pool = torch.cat([F.layer_norm(g_pool, [g_pool.shape[-1]]), F.layer_norm(loc, [300]),
                              F.layer_norm(label, [D1])] + ([label.new_zeros(B, R, kpad)] if kpad else []), dim=2)
            pe = self.pool_embed[0]
            pool = ops.linear(pool, F.pad(pe.weight, (0, kpad)), pe.bias, 1, self._fused_drop_p())   # model.py:384
        if self.has_obj_interact:
            pool = self._obj_interact(pool, enc_key_bias)
        pool = pool.contiguous()
        p_pool = ops.linear(pool, *self._att_proj(self.ctx2pool, 'ctx2pool'))      # MFMA GEMM (model.py:391)
        return self._preamble_finish(segs_feat, sample_idx, fc, pm, pool, p_pool, sim_mat, g_pool)

    def _preamble_finish(self, segs_feat, sample_idx, fc, pm, pool, p_pool, sim_mat, g_pool):
        """fc embedding + the frame half of the preamble (model.py:393-405)."""
        Ft = segs_feat.shape[1]
        fc = self._lin_k32(fc, self.fc_embed[0], act=1, x_padded=fc.shape[-1] != self.fc_embed[0].in_features,
                           p_drop=self._fused_drop_p())
        if self.att_input_mode == 'region':
            # model.py:393,406-409: no frame-wise context (the reference hands the core 1 x 1 dummies it never reads)
            return dict(fc=fc, pool=pool, p_pool=p_pool, conv=None, p_conv=None, g_pool=g_pool, sim_mat_static=sim_mat,
                        pnt_mask=pm, att_input_mode='region', region_attn_mode=self.region_attn_mode)
        # frame-wise context (model.py:393-405)
        # frame embeddings (model.py:393-395) on the MFMA GEMM, straight from the two column blocks of segs_feat
        if not self.training and not torch.is_grad_enabled():
            # inference: the two projections write their column blocks of `c` in place (no concat pass), and BatchNorm1d in
            # eval mode is a per-channel affine of the LAST axis here - applied in place instead of through two [B,H,Ft]
            # transposes around the library kernel (three passes over 0.5 GB at B = 256, Ft = 480; model.py:393-397)
            Hh = self.att_embed[0][0].out_features
            c = torch.empty(segs_feat.shape[0], Ft, 2 * Hh, device=segs_feat.device, dtype=torch.float32)
            c2 = c.view(-1, 2 * Hh)
            ops.gemm_nt(segs_feat[:, :, :2048], self.att_embed[0][0].weight.detach(), self.att_embed[0][0].bias.detach(), 1,
                        out=c2[:, :Hh])
            ops.gemm_nt(segs_feat[:, :, 2048:], self.att_embed[1][0].weight.detach(), self.att_embed[1][0].bias.detach(), 1,
                        out=c2[:, Hh:])
            bn = self.att_embed_aux[0]

            def build_bn(bn=bn):
                scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
                return scale.contiguous(), (bn.bias - bn.running_mean * scale).contiguous()
            scale, shift = self._packed(('bn_affine',), (bn.weight, bn.bias, bn.running_mean, bn.running_var), build_bn)
            ops.affine_relu_rows_(c, scale, shift)
        else:
            c = torch.cat([self._lin(segs_feat[:, :, :2048], self.att_embed[0][0], act=1, p_drop=self._fused_drop_p()),
                           self._lin(segs_feat[:, :, 2048:], self.att_embed[1][0], act=1, p_drop=self._fused_drop_p())],
                          dim=2)
            bn = self.att_embed_aux[0]
            if self.training and bn.momentum is not None and bn.affine and c.is_cuda and c.shape[-1] % 4 == 0:
                # train mode: batch statistics over the B * Ft rows of the [B, Ft, C] layout as it is - column-statistic kernels
                # forward and backward (ops.bn_relu_train), no [B, C, Ft] permute copies around the library BatchNorm
                c = ops.bn_relu_train(c.view(-1, c.shape[-1]), bn).view(c.shape)
            elif not self.training:
                # eval-mode arithmetic under autograd (the gradient goldens): a per-channel affine of the last axis + ReLU
                scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
                c = torch.relu(c * scale + (bn.bias - bn.running_mean * scale))
            else:
                c = self.att_embed_aux(c.permute(0, 2, 1).contiguous()).permute(0, 2, 1).contiguous()
        if not torch.is_grad_enabled():
            # inference: persistent HIP recurrence (one launch per layer instead of ~6 per step/direction)
            rnn = ops.lstm_bidir_2layer if self.t_attn_mode == 'bilstm' else ops.gru_bidir_2layer
            c = rnn(c, self.context_enc, flags=self._flags(), packed=self._packed)
        elif self.t_attn_mode == 'bilstm':
            from . import lstm_fn
            c = lstm_fn.lstm_bidir_2layer_train(c, self.context_enc, flags=self._flags())
        else:
            # training: persistent-kernel forward + hand-scheduled BPTT (gru_fn.py) instead of the library RNN
            from . import gru_fn
            c = gru_fn.gru_bidir_2layer_train(c, self.context_enc, flags=self._flags())
        if not torch.is_grad_enabled() and sample_idx.dtype == torch.int64 and c.is_contiguous():
            conv = ops.zero_rows_outside_window_(c, sample_idx.contiguous())       # in place on the GRU's fresh output
        else:
            t = torch.arange(Ft, device=c.device).view(1, Ft)
            keep = (t >= sample_idx[:, 0:1]) & (t < sample_idx[:, 1:2])           # model.py:303-305
            conv = c.masked_fill(~keep.unsqueeze(-1), 0).contiguous()
        p_conv = ops.linear(conv, *self._att_proj(self.ctx2att, 'ctx2att'))        # MFMA GEMM (model.py:405)
        return dict(fc=fc, pool=pool, p_pool=p_pool, conv=conv, p_conv=p_conv, g_pool=g_pool,
                    sim_mat_static=sim_mat, pnt_mask=pm, att_input_mode=self.att_input_mode,
                    region_attn_mode=self.region_attn_mode)

    @staticmethod
    def _dense_regions(pre):
        """Dense [B,R,.] region features for the consumers that index them directly (beam search, multinomial sampling):
        the compacted preamble keeps them compacted until someone asks."""
        if pre.get('pool') is None:
            pre['pool'], pre['p_pool'] = pre['ci'].expand(pre['pool_c']), pre['ci'].expand(pre['p_pool_c'])
        return pre

    def _decode_params(self):
        c = self.core
        cp = self.core_params()
        return dict(
            embed=self._pad_axis('embed', self.embed[0].weight, 1, self.E_BUILT),
            att_w_ih=cp['att_w_ih'], att_w_hh=c.att_lstm.weight_hh,
            att_b_ih=c.att_lstm.bias_ih, att_b_hh=c.att_lstm.bias_hh,
            lang_w_ih=c.lang_lstm.weight_ih, lang_w_hh=c.lang_lstm.weight_hh,
            lang_b_ih=c.lang_lstm.bias_ih, lang_b_hh=c.lang_lstm.bias_hh,
            att1_h2att_w=cp['a1_w'], att1_h2att_b=cp['a1_b'],
            att1_alpha_w=cp['a1_aw'], att1_alpha_b=c.attention.alpha_net.bias,
            att2_h2att_w=cp['a2_w'], att2_h2att_b=cp['a2_b'],
            logit_w=self.logit.weight, logit_b=self.logit.bias,
            **({} if self.region_attn_mode == 'dp' else
               dict(att2_alpha_w=cp['a2_aw'], att2_alpha_b=c.attention2.alpha_net.bias)))

    # ------------------------------------------------------------------ 'sample' (model.py:492-624)
    def _sample(self, segs_feat, ppls, num, ppls_feat, sample_idx, pnt_mask, opt={}):
        sample_max = opt.get('sample_max', 1)
        beam_size = opt.get('beam_size', 1)
        with torch.no_grad():
            pre = self._preamble(segs_feat, num, ppls, ppls_feat, sample_idx, pnt_mask,
                                 allow_compact=not opt.get('dense_preamble', False))
            P = {k: v.detach() for k, v in self._decode_params().items()}
            if not sample_max:
                from . import sampling
                seq, lps, att2 = sampling.multinomial_decode(self, self._dense_regions(pre), P, opt.get('temperature', 1.0))
            elif beam_size > 1:
                from . import beam
                seq, lps, att2 = beam.beam_decode(self, self._dense_regions(pre), P, beam_size,
                                                   fused_step=opt.get('beam_fused_step', True))
            else:
                seq, lps, att2 = ops.greedy_decode(pre, P, pre['pnt_mask'], self.seq_length, self.unk_idx,
                                                    prof=getattr(self, 'kernel_timer', None), flags=self._flags(),
                                                       att_input_mode=self.att_input_mode, region_attn_mode=self.region_attn_mode)
        if len(self._flags()) + len(self.__dict__.get('_contract_flags', ())) > 4096:   # a caller that never checks must not grow the lists without bound
            self.check_kernel_status()
        return seq, lps, att2, pre['sim_mat_static']

    def sample_pipelined(self, batches, eval_opt={}):
        """Greedy-decode a sequence of independent batches with two HIP streams: the per-segment preamble (fp32-MFMA
        bound) of batch i+1 is enqueued on one stream while the token loop (HBM-bound attention streaming +
        weight-bound LSTMs) of batch i runs on the other, so the two kinds of kernels share the CUs.  `batches`:
        iterable of (segs_feat, ppls, num, ppls_feat, sample_idx, pnt_mask) tuples already on the GPU.  Returns a
        list of (seq, seqLogprobs, att2_weights, sim_mat) — identical values to `_sample` on each batch."""
        assert eval_opt.get('beam_size', 1) == 1 and eval_opt.get('sample_max', 1)
        cur = torch.cuda.current_stream()
        if not hasattr(self, '_streams'):
            self._streams = (torch.cuda.Stream(), torch.cuda.Stream())
        s_pre, s_dec = self._streams
        s_dec.wait_stream(cur)
        outs, keep = [], []          # keep: (preamble tensors, decode-finished event) of the batches still in flight
        P = {k: v.detach() for k, v in self._decode_params().items()}
        trace = eval_opt.get('trace')        # a list: per batch host time stamps + stream events (tools/files_timeline.py)
        import time as _time
        t_prev = _time.perf_counter()
        with torch.no_grad():
            for b in batches:
                tr = None
                if trace is not None:
                    tr = {'asked': t_prev, 'got': _time.perf_counter()}
                    for k in ('pre_start', 'pre_end', 'dec_start', 'dec_end'):
                        tr[k] = torch.cuda.Event(enable_timing=True)
                # `batches` may be a lazy producer (InferenceIngest.batches uploads on the caller's stream while we
                # iterate): order the preamble stream after everything the caller's stream has enqueued so far
                s_pre.wait_stream(cur)
                for t in b:                       # allocated on the caller's stream, read on the side streams: the caching
                    t.record_stream(s_pre)        # allocator must not hand their memory out again before those reads ran
                b[5].record_stream(s_dec)         # (a lazy producer drops its reference as soon as we ask for the next batch)
                with torch.cuda.stream(s_pre):
                    if tr is not None:
                        tr['pre_start'].record(s_pre)
                    pre = self._preamble(b[0], b[2], b[1], b[3], b[4], b[5], allow_compact=True)
                    ev = torch.cuda.Event()
                    ev.record(s_pre)
                    if tr is not None:
                        tr['pre_end'].record(s_pre)
                with torch.cuda.stream(s_dec):
                    s_dec.wait_event(ev)
                    if tr is not None:
                        tr['dec_start'].record(s_dec)
                    seq, lps, att2 = ops.greedy_decode(pre, P, pre['pnt_mask'], self.seq_length, self.unk_idx,
                                                       prof=getattr(self, 'kernel_timer', None), flags=self._flags(),
                                                       att_input_mode=self.att_input_mode, region_attn_mode=self.region_attn_mode)
                    done = torch.cuda.Event()
                    done.record(s_dec)
                    if tr is not None:
                        tr['dec_end'].record(s_dec)
                        tr['enqueued'] = _time.perf_counter()
                # the features stay alive until their token loop has run - but no more than `max_in_flight` batches of them
                # (a long split through a lazy producer would otherwise hold every batch's [B,R,.] tensors until the end):
                # the host waits for the oldest decode before it enqueues further ahead
                keep.append((pre, done))
                while len(keep) > eval_opt.get('max_in_flight', 3):
                    keep.pop(0)[1].synchronize()
                # (a whole split through a lazy producer: callers that need neither tensor say so - sim_mat_static is
                # [B, 433, R], 1.7 MB per segment, att2 [B, L, R] 80 KB - instead of holding them for every batch until the end)
                outs.append((seq, lps, att2 if eval_opt.get('keep_att2', True) else None,
                             pre['sim_mat_static'] if eval_opt.get('keep_sim_mat', True) else None))
                if tr is not None:
                    tr['throttled'] = t_prev = _time.perf_counter()
                    trace.append(tr)
        cur.wait_stream(s_pre)
        cur.wait_stream(s_dec)
        for o in outs:                               # allocated on the side streams, consumed on the caller's
            for t in o:
                if t is not None:
                    t.record_stream(cur)
        self._pipeline_keepalive = keep              # released on the next call, after the streams were joined
        counts = self.kernel_status_counts()
        if counts is not None:
            bad, contract = counts.tolist()
            if (bad or contract) and isinstance(batches, (list, tuple)):
                # some batch broke the zero-row loader contract, or a persistent kernel timed out on a shared GPU: compute,
                # don't raise (see _sample_checked) - every batch again, one by one (a lazy producer that recycles its
                # staging buffers cannot be replayed: that case raises below)
                if bad:
                    ops.disable_persistent_kernels(bad)
                opt2 = dict(eval_opt, dense_preamble=True) if contract else eval_opt
                outs = [self._sample(b[0], b[1], b[2], b[3], b[4], b[5], opt2) for b in batches]
                c2 = self.kernel_status_counts()
                bad, contract = (0 if c2 is None else c2.tolist()[0]), 0
            self.raise_for_status(bad, contract)
        return outs

    # ------------------------------------------------------------------ 'MLE' / 'GRD' (model.py:283-489)
    def _forward_train(self, segs_feat, input_seq, gt_seq, ppls, gt_boxes, mask_boxes, num, ppls_feat,
                       frm_mask, sample_idx, pnt_mask, eval_obj_ground):
        from . import train_step
        return train_step.forward_train(self, segs_feat, input_seq, gt_seq, ppls, gt_boxes, mask_boxes, num,
                                        ppls_feat, frm_mask, sample_idx, pnt_mask, eval_obj_ground)


def attended_region_indices(att2_weights, num_sampled_frm, num_prop_per_frm):
    """main.py:364-365: per-frame argmax over the P proposals of each of the T frames of the masked attention
    logits returned by `'sample'` -> int64 [B,L,T] (the "attended region indices" of the parity contract)."""
    B, L = att2_weights.shape[0], att2_weights.shape[1]
    return att2_weights.view(B, L, num_sampled_frm, num_prop_per_frm).max(dim=-1)[1]


class AttModel(TopDownModel):
    """Alias kept for drivers that import `misc.model.AttModel` semantics."""
