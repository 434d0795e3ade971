"""'MLE' / 'GRD' drivers (AttModel._forward, model.py:283-489) on the HIP path.

The teacher-forced token loop runs through `decoder_loop` (HIP LSTM + attention kernels per step, no
host synchronisation inside the loop: the early `break` of model.py:425 is resolved once up front
from the ground-truth sequence), box targets for all steps come from one kernel each, and the losses
are fused log-softmax reductions.
"""
import os

import torch
import torch.nn.functional as F

from . import ops, train_compact


def _seq_cnt(seq, L):
    """Number of loop iterations model.py:421-426 executes: stops at the first i>=1 whose column is all 0."""
    col_zero = (seq[:, 1:L] == 0).all(dim=0)           # columns i = 1..L-1
    nz = torch.nonzero(col_zero)
    return int(nz[0]) + 1 if nz.numel() else L           # one host sync per batch (the reference syncs per step)


def decoder_loop(model, pre, xt_all, att_masks, pnt_masks):
    """Lc teacher-forced steps of TopDownCore.forward (AttModel.py:134-164).

    xt_all [B,Lc,E] embedded inputs; att_masks u8 [B,R+1] (same every step); pnt_masks u8 [B,Lc,R+1] or
    [B,R+1].  Returns h_lang_all [B,Lc,H], att2_weights [B,Lc,R] (masked pre-softmax logits)."""
    from . import decoder_fn
    return decoder_fn.decoder_loop(model, pre, xt_all, att_masks, pnt_masks)


def train_compact_enabled(model):
    """The compacted training layout is on for this model: its `train_compact` attribute (train.Trainer(compact_rows=...))
    or, when that is unset, GVD_TRAIN_COMPACT=1 - and no step has met inputs that break the zero-row premise yet."""
    want = getattr(model, 'train_compact', None)
    if want is None:
        want = os.environ.get('GVD_TRAIN_COMPACT', '0') == '1'
    return bool(want) and not getattr(model, '_train_compact_off', False)


def forward_train(model, segs_feat, input_seq, gt_seq, ppls, gt_boxes, mask_boxes, num, ppls_feat, frm_mask,
                  sample_idx, pnt_mask, eval_obj_ground):
    B, R = segs_feat.shape[0], ppls.shape[1]
    L, V = model.seq_length, model.vocab_size
    T, P = model.num_sampled_frm, model.num_prop_per_frm
    dev = segs_feat.device
    seq = gt_seq[:, 0, :]
    seq = torch.cat([torch.zeros(B, 1, dtype=seq.dtype, device=dev), seq], 1)            # model.py:285-286
    input_seq = input_seq.view(-1, input_seq.shape[2], input_seq.shape[3])
    key_bias = None
    if not eval_obj_ground and torch.is_grad_enabled() and train_compact_enabled(model) and frm_mask.dim() == 3:
        # masked-proposal compaction of the training step (train_compact.py; OFF by default: see its status note): the
        # step runs on [valid rows | one weighted representative masked row | pads] per segment.  The premise - masked
        # rows are zero rows, dataloader_anet.py:343-344 - is checked on the device; train.Trainer re-runs a step whose
        # inputs break it on the full row set
        pm0 = (pnt_mask if pnt_mask.dtype == torch.uint8 else pnt_mask.to(torch.uint8)).contiguous()
        c = train_compact.compact_regions(ppls, ppls_feat, pm0, frm_mask)
        if c is not None:
            flag = torch.zeros(1, dtype=torch.int32, device=pm0.device)
            ops.check_masked_rows_zero(ppls_feat.contiguous(), pm0, flag)
            ops.check_masked_rows_zero(ppls.contiguous(), pm0, flag)
            model.__dict__.setdefault('_contract_flags', []).append(flag)
            ppls, ppls_feat, pnt_mask, frm_mask, key_bias = c['ppls'], c['ppls_feat'], c['pnt_mask'], c['frm_mask'], c['key_bias']
            R = c['Rc']
    pre = model._preamble(segs_feat, num, ppls, ppls_feat, sample_idx, pnt_mask, enc_key_bias=key_bias)
    pm = pre['pnt_mask']
    fm = frm_mask if frm_mask.dtype == torch.uint8 else frm_mask.to(torch.uint8)
    mb = mask_boxes if mask_boxes.dtype == torch.uint8 else mask_boxes.to(torch.uint8)

    # box targets (model.py:317-318,345): IoU + sim target in one kernel
    overlaps, sim_target = ops.iou_targets(ppls.detach().float(), gt_boxes.detach().float(), fm, pm)
    sim_mat = pre['sim_mat_static']
    sim_mask = sim_target > 0
    cls_loss = cls_pred = None
    if not model.test_mode:
        if not eval_obj_ground:
            # model.py:348-350 (BCE against ones over sim_target > 0): fused gather + log + masked mean (T2)
            cls_loss = ops.cls_loss(sim_mat, sim_target)
        else:
            tgt = torch.masked_select(sim_target, sim_mask)
            prd = torch.masked_select(sim_mat.max(dim=1)[1].unsqueeze(1).expand_as(sim_target), sim_mask)
            cls_pred = torch.stack([tgt, prd], dim=1)
    else:
        cls_pred = 0

    Lc = _seq_cnt(seq, L)
    xt_all = model.padded_embedding(model._drop(F.relu(model.embed[0](seq[:, :Lc]))))   # model.py:428
    if not eval_obj_ground:
        roi_labels, frm_masks = ops.step_targets(overlaps, mb, fm, pm, Lc)              # model.py:431-440
        h_all, att2_weights = decoder_loop(model, pre, xt_all, pm, frm_masks)
    else:
        h_all, att2_weights = decoder_loop(model, pre, xt_all, pm, pm)
    h_all = model._drop(h_all)                                                           # AttModel.py:161

    # grounding (model.py:469-480)
    xt_clamp = torch.clamp(input_seq[:, 1:Lc + 1, 0] - V, min=0)
    xt_vis = model._drop(F.relu(model.vis_embed[0](xt_clamp))).contiguous()
    cls_bias = model.vis_classifiers_bias[xt_clamp].contiguous()                        # [B,Lc]
    if eval_obj_ground:
        ground = ops.grounder(xt_vis, pre['g_pool'], pm[:, 1:], mbias=cls_bias, rowbias=att2_weights)
        return (cls_pred, att2_weights.view(B, Lc, T, P).max(dim=-1)[1],
                ground.view(B, Lc, T, P).max(dim=-1)[1])
    ground = ops.grounder(xt_vis, pre['g_pool'], frm_masks[:, :, 1:], mbias=cls_bias, rowbias=att2_weights)

    # losses (utils.py:122-152)
    logits = ops.linear(h_all.reshape(B * Lc, -1), model.logit.weight, model.logit.bias)
    target = seq[:, 1:Lc + 1].contiguous()
    logp_t = ops.nll_gather(logits, target.reshape(-1))                                 # log p[target]
    tmask = torch.cat([torch.ones(B, 1, dtype=torch.bool, device=dev), target[:, :-1] > 0], 1).reshape(-1)
    lm_loss = -(logp_t * tmask.to(logp_t.dtype)).sum() / tmask.sum()
    att2_loss = ops.masked_lsm(att2_weights, roi_labels)
    ground_loss = ops.masked_lsm(ground, roi_labels)
    if len(model._flags()) + len(model.__dict__.get('_contract_flags', [])) > 4096:
        model.check_kernel_status()      # a caller that never checks must not grow the lists without bound
    return lm_loss.unsqueeze(0), att2_loss.unsqueeze(0), ground_loss.unsqueeze(0), cls_loss.unsqueeze(0)
