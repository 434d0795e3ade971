"""Driver-side contract of main.py around the model: what `train()` / `eval()` do with the
model's inputs and outputs, without the dataset / metric submodules that are out of scope.

  decode_sequence        utils.py:90-106   token ids -> sentences (stops at the first END = 0)
  grounding_boxes        main.py:364-368   per-frame attended proposal boxes for every generated word
  collect_predictions    main.py:370-400   the `predictions` / `grd_output` dictionaries main.py dumps as JSON
  save_checkpoint / load_checkpoint   main.py:622-652, 702-743   model.pth / model-best.pth + infos_<id>.pkl +
                         histories_<id>.pkl, interchangeable with the reference (same state_dict keys and shapes)
  train_epoch            main.py:197-311   loss bookkeeping around Trainer.step
  eval_split             main.py:313-452   inference loop over the feature-ingest pipeline + the two result JSON files
  collect_gt_grounding / eval_grounding   main.py:87-194   'GRD' on the ground-truth sentences -> attn-gt-sent-results-*.json,
                         grd-gt-sent-results-*.json and the per-class classification accuracy
  run_epochs             main.py:678-743   epoch loop: LR decay schedule, train, validate every k epochs, model.pth /
                         model-best.pth + infos/histories on improving validation score
"""
import os
import pickle
from collections import defaultdict

import torch

from .hip import GvdHipError


def decode_sequence(itow, seq):
    """utils.decode_sequence (utils.py:90-106): words joined by ' ', stop at token 0; itow maps str(id) -> word."""
    out = []
    for row in seq.tolist():
        txt = ''
        for j, ix in enumerate(row):
            if j >= 1:
                txt = txt + ' '
            if ix == 0:
                break
            txt = txt + itow[str(ix)]
        out.append(txt)
    return out


def grounding_boxes(att2_weights, ppls, num_sampled_frm, num_prop_per_frm):
    """main.py:364-368.  att2_weights [B,L,R] (masked logits from 'sample'), ppls [B,R,7] ->
    (att2_ind i64 [B,L,T], obj_bbox_att2 f32 [B,L,T,7]): for every word and frame the attended proposal."""
    B, L = att2_weights.shape[0], att2_weights.shape[1]
    T, P = num_sampled_frm, num_prop_per_frm
    att2_ind = att2_weights.view(B, L, T, P).max(dim=-1)[1]
    boxes = torch.gather(ppls.view(-1, T, P, ppls.shape[-1]).permute(0, 2, 1, 3).contiguous(), 1,
                         att2_ind.unsqueeze(-1).expand(B, L, T, ppls.shape[-1]))
    return att2_ind, boxes


def collect_predictions(seq, seg_ids, itow, timestamps=None, att2_weights=None, ppls=None, opt=None,
                        wtol=None, lemma_det_dict=None, itod=None, predictions=None, grd_output=None):
    """main.py:370-400 for one batch: appends {'sentence', 'timestamp'} per segment to predictions[vid] and, when
    grounding evaluation is on (att2_weights given), {'clss','idx_in_sent','bbox_for_all_frames'} to
    grd_output[vid][seg].  seg_ids: '<vid>_segment_<k>' strings."""
    predictions = defaultdict(list) if predictions is None else predictions
    grd_output = defaultdict(dict) if grd_output is None else grd_output
    seq_cpu = seq.cpu()
    if att2_weights is not None:
        _, boxes = grounding_boxes(att2_weights, ppls, opt.num_sampled_frm, opt.num_prop_per_frm)
        boxes = boxes.cpu()
        for i in range(seq_cpu.shape[0]):
            vid_id, seg_idx = seg_ids[i].split('_segment_')
            seg_idx = str(int(seg_idx))
            tmp = {'clss': [], 'idx_in_sent': [], 'bbox_for_all_frames': []}
            for j in range(seq_cpu.shape[1]):
                tok = int(seq_cpu[i, j])
                if tok == 0:
                    break
                lemma = wtol[itow[str(tok)]]
                if lemma in lemma_det_dict:
                    tmp['bbox_for_all_frames'].append(boxes[i, j, :, :4].tolist())
                    tmp['clss'].append(itod[lemma_det_dict[lemma]])
                    tmp['idx_in_sent'].append(j)
            grd_output[vid_id][seg_idx] = tmp
    for k, sent in enumerate(decode_sequence(itow, seq_cpu)):
        vid_idx, seg_idx = seg_ids[k].split('_segment_')
        seg_idx = str(int(seg_idx))
        entry = {'sentence': sent}
        if timestamps is not None:
            entry['timestamp'] = [round(t, 2) for t in timestamps[vid_idx][seg_idx]]
        predictions[vid_idx].append(entry)
    return predictions, grd_output


def save_checkpoint(model, opt, checkpoint_path, infos=None, histories=None, best=False, itow=None):
    """main.py:702-743: model.pth (+ model-best.pth), infos_<id>.pkl (+ -best), histories_<id>.pkl.
    The optimizer state is not saved (main.py:715-716 keeps it commented out)."""
    os.makedirs(checkpoint_path, exist_ok=True)
    module = model.module if hasattr(model, 'module') else model
    sd = {k: v.detach().cpu() for k, v in module.state_dict().items()}
    infos = dict(infos or {})
    infos.setdefault('iter', 0)
    infos.setdefault('epoch', 0)
    infos.setdefault('best_val_score', None)
    infos['opt'] = opt
    infos['vocab'] = itow
    torch.save(sd, os.path.join(checkpoint_path, 'model.pth'))
    with open(os.path.join(checkpoint_path, 'infos_' + opt.id + '.pkl'), 'wb') as f:
        pickle.dump(infos, f)
    with open(os.path.join(checkpoint_path, 'histories_' + opt.id + '.pkl'), 'wb') as f:
        pickle.dump(histories or {}, f)
    if best:
        torch.save(sd, os.path.join(checkpoint_path, 'model-best.pth'))
        with open(os.path.join(checkpoint_path, 'infos_' + opt.id + '-best.pkl'), 'wb') as f:
            pickle.dump(infos, f)


def load_checkpoint(model, start_from, run_id, load_best_score=0, map_location='cpu'):
    """main.py:622-652: returns (infos, histories); the model gets the saved state_dict (strict)."""
    tag = '-best' if load_best_score == 1 else ''
    model_path = os.path.join(start_from, 'model-best.pth' if load_best_score == 1 else 'model.pth')
    with open(os.path.join(start_from, 'infos_' + run_id + tag + '.pkl'), 'rb') as f:
        infos = pickle.load(f, encoding='latin1')
    module = model.module if hasattr(model, 'module') else model
    module.load_state_dict(torch.load(model_path, map_location=map_location))
    histories = {}
    hp = os.path.join(start_from, 'histories_' + run_id + '.pkl')
    if os.path.isfile(hp):
        with open(hp, 'rb') as f:
            histories = pickle.load(f, encoding='latin1')
    return infos, histories


def train_epoch(trainer, batches, opt, log=None):
    """main.py:197-311: one pass over `batches` (iterables of the 11 positional model inputs); returns the running
    means main.py prints (train loss, lm, att2, ground, cls).  The number of optimisation steps of the pass is left in
    `trainer.steps_last_epoch` (run_epochs advances infos['iter'] by it)."""
    trainer.model.train()
    sums = torch.zeros(5)
    n = 0
    for step, args in enumerate(batches):
        losses = trainer.step(args).float().cpu()         # lm, att2, ground, cls (each already / n_replicas = 1)
        total = losses[0] + opt.w_att2 * losses[1] + opt.w_grd * losses[2] + opt.w_cls * losses[3]
        sums += torch.cat([total.view(1), losses[0:1], opt.w_att2 * losses[1:2], opt.w_grd * losses[2:3],
                           opt.w_cls * losses[3:4]])
        n += 1
        if log is not None and step % max(getattr(opt, 'disp_interval', 100), 1) == 0:
            log('step %d: train_loss %.4f (lm %.4f att2 %.4f grd %.4f cls %.4f)' % ((step,) + tuple((sums / n).tolist())))
    trainer.steps_last_epoch = n
    return (sums / max(n, 1)).tolist()


def eval_split(model, ingest_pipeline, records, batch_size, itow, opt, eval_opt=None, timestamps=None, wtol=None,
               lemma_det_dict=None, itod=None, out_dir=None, val_split='validation', pipelined=False):
    """Inference half of main.eval (main.py:313-452) over the ingest pipeline: for every batch of segment records
    `model(..., 'sample', eval_opt)` -> sentences (+ per-word grounding boxes when `lemma_det_dict` is given), collected
    into the `predictions` / `grd_output` dictionaries and, with `out_dir`, written as the two JSON files main.py hands
    to its (out-of-scope) evaluators: densecap-<split>-<id>.json (main.py:418-424) and
    attn-gen-sent-results-<split>-<id>.json (main.py:446-449).
    pipelined (greedy decode only): the batches go through TopDownModel.sample_pipelined - file reads of batch i+2 (reader
    threads), upload of batch i+1 (copy stream), preamble of batch i+1 and token loop of batch i (two HIP streams) all in
    flight together, the sentences collected at the end; same results as the batch-by-batch loop."""
    import json
    eval_opt = eval_opt or {'sample_max': 1, 'beam_size': getattr(opt, 'beam_size', 1), 'inference_mode': True}
    predictions, grd_output = defaultdict(list), defaultdict(dict)
    grounding = lemma_det_dict is not None
    model.eval()
    if pipelined and eval_opt.get('beam_size', 1) == 1 and eval_opt.get('sample_max', 1):
        chunks, ppls_of = [], []

        def produce():
            for chunk, t in ingest_pipeline.batches(records, batch_size):
                chunks.append(chunk)
                ppls_of.append(t['ppls'])
                yield (t['segs_feat'], t['ppls'], t['num'], t['ppls_feat'], t['sample_idx'], t['pnt_mask'])
        try:
            # (the sentences need the ids only; the attention logits only when grounding boxes are asked for: nothing else of a
            # batch stays on the device until the end of the split)
            outs = model.sample_pipelined(produce(), dict(eval_opt, keep_sim_mat=False, keep_att2=grounding))
            for chunk, ppls, (seq, _, att2_weights, _) in zip(chunks, ppls_of, outs):
                collect_predictions(seq, [r['seg_id'] for r in chunk], itow, timestamps=timestamps,
                                    att2_weights=att2_weights if grounding else None, ppls=ppls, opt=opt, wtol=wtol,
                                    lemma_det_dict=lemma_det_dict, itod=itod, predictions=predictions, grd_output=grd_output)
            records = ()                              # (done: the loop below has nothing left)
        except GvdHipError:
            # a lazily produced batch cannot be replayed inside sample_pipelined (its staging buffers were recycled): a batch
            # that broke the loader's zero-row contract, or a persistent-kernel barrier timeout, lands here - the records CAN be
            # read again: the batch-by-batch loop below recomputes every batch through forward(..., 'sample'), which handles both
            predictions.clear()
            grd_output.clear()
    with torch.no_grad():
        for chunk, t in ingest_pipeline.batches(records, batch_size):
            dummy = t['ppls'].new_zeros(t['ppls'].shape[0]).byte()                     # main.py:353
            seq, att2_weights, sim_mat = model(t['segs_feat'], dummy, dummy, t['num'], t['ppls'], dummy, dummy,
                                               t['ppls_feat'], dummy, t['sample_idx'], t['pnt_mask'], 'sample', eval_opt)
            collect_predictions(seq, [r['seg_id'] for r in chunk], itow, timestamps=timestamps,
                                att2_weights=att2_weights if grounding else None, ppls=t['ppls'], opt=opt, wtol=wtol,
                                lemma_det_dict=lemma_det_dict, itod=itod, predictions=predictions, grd_output=grd_output)
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, 'densecap-%s-%s.json' % (val_split, opt.id)), 'w') as f:
            json.dump({'version': 'VERSION 1.0', 'results': predictions,
                       'external_data': {'used': 'true', 'details': 'Visual Genome for Faster R-CNN pre-training'}}, f)
        if grounding:
            with open(os.path.join(out_dir, 'attn-gen-sent-results-%s-%s.json' % (val_split, opt.id)), 'w') as f:
                json.dump({'results': grd_output, 'eval_mode': 'gen',
                           'external_data': {'used': True, 'details': 'Object detector pre-trained on Visual Genome on '
                                                                      'object detection task.'}}, f)
    return predictions, grd_output


def collect_gt_grounding(att2_ind, grd_ind, input_seqs, ppls, seg_ids, opt, itod, att2_output=None, grd_output=None,
                         vocab_in_split=None):
    """main.py:128-153 for one 'GRD' batch: the boxes of the attended (att2_ind) and grounded (grd_ind) proposals of every
    grounded GT word (`input_seqs[:,0,1:,0] > vocab_size`), per frame, keyed like the reference's result files.
    att2_ind / grd_ind: i64 [B,Lc,T] (model 'GRD' outputs); input_seqs [B,1,L+1,4]; ppls [B,R,7]."""
    att2_output = defaultdict(dict) if att2_output is None else att2_output
    grd_output = defaultdict(dict) if grd_output is None else grd_output
    vocab_in_split = set() if vocab_in_split is None else vocab_in_split
    T, P = opt.num_sampled_frm, opt.num_prop_per_frm
    obj_mask = (input_seqs[:, 0, 1:, 0] > opt.vocab_size).cpu()                     # main.py:129
    by_frame = ppls.view(-1, T, P, ppls.shape[-1]).permute(0, 2, 1, 3).contiguous()  # [B,P,T,7]

    def boxes(ind):
        B, Lc = ind.shape[0], ind.shape[1]
        return torch.gather(by_frame, 1, ind.unsqueeze(-1).expand(B, Lc, T, ppls.shape[-1])).cpu()
    b_att2, b_grd = boxes(att2_ind), boxes(grd_ind)
    seq_cpu = input_seqs.cpu()
    for i in range(obj_mask.shape[0]):
        vid_id, seg_idx = seg_ids[i].split('_segment_')
        seg_idx = str(int(seg_idx))
        r_att2 = {'clss': [], 'idx_in_sent': [], 'bbox_for_all_frames': []}
        r_grd = {'clss': [], 'idx_in_sent': [], 'bbox_for_all_frames': []}
        for j in range(min(obj_mask.shape[1], b_att2.shape[1])):
            if obj_mask[i, j]:
                cls_name = itod[int(seq_cpu[i, 0, j + 1, 0]) - opt.vocab_size]
                vocab_in_split.add(cls_name)
                for r, bx in ((r_att2, b_att2), (r_grd, b_grd)):
                    r['clss'].append(cls_name)
                    r['idx_in_sent'].append(j)
                    r['bbox_for_all_frames'].append(bx[i, j, :, :4].tolist())
        att2_output[vid_id][seg_idx] = r_att2
        grd_output[vid_id][seg_idx] = r_grd
    return att2_output, grd_output, vocab_in_split


def class_accuracy(cls_pred, vocab_in_split):
    """main.py:166-171: mean over the classes of the split of the per-class hit rate of the region classifier.
    cls_pred: i64 [N,2] rows (GT class, predicted class) concatenated over the split ('GRD' first output)."""
    score = defaultdict(list)
    hit = (cls_pred[:, 0] == cls_pred[:, 1]).long()
    for c, h in zip(cls_pred[:, 0].tolist(), hit.tolist()):
        score[c].append(h)
    return sum(sum(h) * 1.0 / len(h) for h in score.values()) * 1.0 / max(len(vocab_in_split), 1), len(score)


def eval_grounding(model, batches, opt, itod, out_dir=None, val_split='validation', evaluator=None):
    """main.eval_grounding (main.py:87-194): 'GRD' over the ground-truth sentences of a split.  `batches` yields
    (seg_ids, the 11 positional model inputs); mask_boxes may be the reference's dummy (main.py:122).  Writes
    attn-gt-sent-results-<split>-<id>.json and grd-gt-sent-results-<split>-<id>.json (main.py:157-163) and returns
    (attn_accu, grd_accu, cls_accu); the box-accuracy numbers come from `evaluator(attn_file, grd_file)` — the
    reference's ANetGrdEval submodule is outside this repository — and are 0 without one / in test mode."""
    import json
    att2_output, grd_output, vocab = defaultdict(dict), defaultdict(dict), set()
    cls_pred = []
    model.eval()
    with torch.no_grad():
        for seg_ids, args in batches:
            cp, att2_ind, grd_ind = model(*args, 'GRD')
            collect_gt_grounding(att2_ind, grd_ind, args[1], args[4], seg_ids, opt, itod, att2_output, grd_output, vocab)
            cls_pred.append(cp.cpu())
    attn_file = grd_file = None
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        ext = {'used': True, 'details': 'Object detector pre-trained on Visual Genome on object detection task.'}
        attn_file = os.path.join(out_dir, 'attn-gt-sent-results-%s-%s.json' % (val_split, opt.id))
        with open(attn_file, 'w') as f:
            json.dump({'results': att2_output, 'eval_mode': 'GT', 'external_data': ext}, f)
        grd_file = os.path.join(out_dir, 'grd-gt-sent-results-%s-%s.json' % (val_split, opt.id))
        with open(grd_file, 'w') as f:
            json.dump({'results': grd_output, 'eval_mode': 'GT', 'external_data': ext}, f)
    if getattr(opt, 'test_mode', False):
        return 0, 0, 0                                                  # main.py:187-194
    cls_accu, _ = class_accuracy(torch.cat(cls_pred, 0), vocab) if cls_pred else (0.0, 0)
    attn_accu = grd_accu = 0.0
    if evaluator is not None and attn_file is not None:
        attn_accu, grd_accu = evaluator(attn_file, grd_file)
    return attn_accu, grd_accu, cls_accu


def set_lr(optimizer, decay_factor):
    """utils.set_lr (utils.py:155-157)."""
    for group in optimizer.param_groups:
        group['lr'] = group['lr'] * decay_factor


def run_epochs(trainer, opt, train_batches, validate, checkpoint_path=None, infos=None, histories=None, itow=None,
               log=print):
    """The epoch loop of main.py:678-743 around `Trainer` (the reference's `train()` / `eval()` become the callables):
      * learning-rate decay: for epoch > learning_rate_decay_start >= 0, every learning_rate_decay_every epochs every
        group's lr and opt.learning_rate are multiplied by learning_rate_decay_rate (main.py:679-683);
      * `train_batches(epoch)` yields the 11-tuples of one epoch (skipped under opt.inference_only);
      * every val_every_epoch epochs `validate(epoch)` returns the language stats dict; its 'CIDEr' is the model
        selection score: model.pth + infos/histories are written every validation, model-best.pth when it improves
        (main.py:700-743).
    Conscious deviation: the reference never increments its `iteration` (main.py:309-311,646), so its
    infos['iter'] is always the resumed value; here infos['iter'] counts the optimisation steps run so far, so that a
    resumed run knows its position.  lr / loss histories stay keyed by epoch (the reference's iteration-keyed entries all
    collapse onto the single key `iteration`).
    Returns (infos, histories) exactly as they are pickled."""
    infos = dict(infos or {})
    histories = dict(histories or {})
    best = infos.get('best_val_score', None)
    start_epoch = infos.get('epoch', 0)
    iteration = infos.get('iter', 0)
    val_hist = histories.setdefault('val_result_history', {})
    histories.setdefault('loss_history', {})
    lr_hist = histories.setdefault('lr_history', {})
    histories.setdefault('ss_prob_history', {})
    for epoch in range(start_epoch, opt.max_epochs):
        if epoch > opt.learning_rate_decay_start and opt.learning_rate_decay_start >= 0:
            if (epoch - opt.learning_rate_decay_start) % opt.learning_rate_decay_every == 0:
                set_lr(trainer.optimizer, opt.learning_rate_decay_rate)
                opt.learning_rate = opt.learning_rate * opt.learning_rate_decay_rate
        lr_hist[epoch] = opt.learning_rate
        if not getattr(opt, 'inference_only', False):
            means = train_epoch(trainer, train_batches(epoch), opt, log=log)
            iteration += getattr(trainer, 'steps_last_epoch', 0)
            histories['loss_history'][epoch] = means[0]
        if epoch % opt.val_every_epoch == 0:
            with torch.no_grad():
                lang_stats = validate(epoch)
            if getattr(opt, 'inference_only', False):
                break
            score = lang_stats['CIDEr']
            val_hist[epoch] = lang_stats
            best_flag = best is None or score > best
            if best_flag:
                best = score
            infos.update(iter=iteration, epoch=epoch, best_val_score=best)
            if checkpoint_path is not None:
                save_checkpoint(trainer.model, opt, checkpoint_path, infos=infos, histories=histories, best=best_flag,
                                itow=itow)
                if log:
                    log('model saved to %s%s' % (os.path.join(checkpoint_path, 'model.pth'),
                                                  ' (best CIDEr %.3f)' % best if best_flag else ''))
    return infos, histories
