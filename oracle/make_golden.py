"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference) on the
seeded cases of oracle/cases.py.  Build-container only; the fixtures are committed so that the GPU
box (no /root/reference) can check both the oracle and the HIP path against reference outputs.

    python -m oracle.make_golden [case ...]        # from the repo root

What is stored per case (all produced by the reference model, eval mode, CPU fp32, torch 2.10):
  sample: seq i64[B,L], seqLogprobs f32[B,L], att_idx i16[B,L,T] (main.py:364-365 per-frame argmax),
          att2_weights f32[B,L,R] (only B<=4), sim_sub f32[B,D1,11] (every 97th region of sim_mat)
  MLE:    losses f32[4] (lm, att2, ground, cls); grad_norms (per-parameter L2 norm of the gradient of
          lm + w_att2*att2 + w_grd*ground + w_cls*cls, cases.GRAD_WEIGHTS) and grad_proj f64[n_params,8]: seeded random
          projections of every parameter's gradient (cases.grad_projections) - pins the DIRECTION of the gradient, not
          only its magnitude.  `bn_train` cases run the reference in train mode with every dropout ratio 0 (BatchNorm
          batch statistics) and also store the updated running statistics
  beam:   seq i64[B,L], seqLogprobs f32[B,L], att2 i32[B,L] (global argmax-over-R region index per step) from the
          reference's own beam_search run under ref_harness.beam_shim (the unshimmed reference raises)
  step:   one main.train optimisation step (loss assembly, clip 0.1, Adam with the fc7/vis_embed lr x0.1 groups): per-parameter
          update norms, Adam first-moment norms, the pre-clip total gradient norm, the losses
  ingest: the eleven model inputs produced by the reference's REAL DataLoader.__getitem__ + main.py's trimming on a seeded
          synthetic dataset (small tensors whole, feature tensors as exact checksums)
  GRD:    cls_pred i64[N,2], att2_ind i16[B,Lc,T], grd_ind i16[B,Lc,T]
plus weight/input fingerprints (exact integer checksums of the raw bits) so a consumer can prove it regenerated the
identical weights and inputs from the seeds.
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cases, ref_harness  # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def run_ingest_case(name):
    """The reference's real dataloader on the synthetic files of the case -> small tensors whole, large ones as exact
    integer checksums of their bits."""
    import tempfile
    from oracle import ref_dataloader_harness as RH
    t0 = time.time()
    with tempfile.TemporaryDirectory() as root:
        opt, vocab, fr, sr, recs = cases.build_ingest_case(name, root)
        ds = RH.build_reference_dataset(recs, vocab, fr, sr, opt)
        ref = RH.reference_batch(ds, list(range(len(recs))), train=True)
    out = dict(torch_version=np.array(torch.__version__), n_records=np.int64(len(recs)))
    for k in cases.INGEST_KEYS:
        out['shape_' + k] = np.array(ref[k].shape, dtype=np.int64)
        out['fp_' + k] = np.int64(cases._bits_checksum(ref[k]))
        if k in cases.INGEST_SMALL:
            out[k] = ref[k].numpy()
    np.savez_compressed(os.path.join(GOLDEN_DIR, name + '.npz'), **out)
    print('%-40s %6.1fs  %d records' % (name, time.time() - t0, len(recs)))


def run_case(name):
    import importlib
    pkg = importlib.import_module('grounded-video-description_amd')
    spec = cases.CASES[name]
    if spec['mode'] == 'ingest':
        return run_ingest_case(name)
    opt, sd, inp = cases.build_case(name)
    need_grad = spec['mode'] in ('MLE', 'step', 'dp', 'traj')
    ref = ref_harness.build_reference_model(opt, sd, need_grad=need_grad).eval()
    if spec.get('bn_train'):
        cases.zero_dropout(ref).train()
    args = pkg.synth.as_args(inp)
    out = dict(weight_fp=np.int64(cases.weight_fingerprint(sd)),
               input_fp=np.int64(cases.input_fingerprint(inp)),
               torch_version=np.array(torch.__version__))
    t0 = time.time()
    if spec['mode'] == 'sample':
        with torch.no_grad():
            seq, lps, att2, sim = ref._sample(inp['segs_feat'], inp['ppls'], inp['num'], inp['ppls_feat'],
                                              inp['sample_idx'], inp['pnt_mask'],
                                              {'sample_max': 1, 'beam_size': 1})
        B, L = seq.shape
        idx = att2.view(B, L, opt.num_sampled_frm, opt.num_prop_per_frm).max(dim=-1)[1]
        out.update(seq=seq.numpy(), seqLogprobs=lps.numpy(), att_idx=idx.numpy().astype(np.int16))
        out['sim_sub'] = cases.sim_sub(sim).contiguous().numpy()
        if B <= 4:
            out['att2_weights'] = att2.numpy()
        if 'slice' in spec:
            a, b = spec['slice']
            with torch.no_grad():
                sseq, slps, satt2, _ = ref._sample(*[inp[k][a:b].contiguous() for k in (
                    'segs_feat', 'ppls', 'num', 'ppls_feat', 'sample_idx', 'pnt_mask')], {'sample_max': 1, 'beam_size': 1})
            sidx = satt2.view(b - a, L, opt.num_sampled_frm, opt.num_prop_per_frm).max(dim=-1)[1]
            out.update(slice_seq=sseq.numpy(), slice_att_idx=sidx.numpy().astype(np.int16),
                       slice_seqLogprobs=slps.numpy())
    elif spec['mode'] == 'beam':
        seq, lps, att2 = ref_harness.reference_beam_sample(ref, inp, spec['K'])
        out.update(seq=seq.numpy(), seqLogprobs=lps.numpy(), att2=att2.numpy().astype(np.int32))
    elif spec['mode'] == 'MLE':
        if need_grad:
            lm, a2, gl, cl = ref(*args, 'MLE')
            w = cases.GRAD_WEIGHTS
            loss = lm.sum() + w['w_att2'] * a2.sum() + w['w_grd'] * gl.sum() + w['w_cls'] * cl.sum()
            ref.zero_grad()
            loss.backward()
            names, norms, projs = [], [], []
            for n, p in ref.named_parameters():
                if p.grad is not None:
                    names.append(n)
                    norms.append(float(p.grad.double().norm()))
                    projs.append(cases.grad_projections(n, p.grad))
            out.update(grad_names=np.array(names), grad_norms=np.array(norms, dtype=np.float64),
                       grad_proj=np.array(projs, dtype=np.float64))
            if spec.get('f64_grads'):
                # the fp64 value of the same gradient (the oracle in double precision; tools/reference_grad_noise.py): lets a
                # consumer tell the reference's own fp32 rounding (a ReLU unit on the other side of zero) from an error
                from oracle import gvd_oracle as O
                torch.set_default_dtype(torch.float64)
                try:
                    W = {k: (v.double().clone().requires_grad_('running' not in k) if v.is_floating_point() else v)
                         for k, v in sd.items()}
                    a64 = [inp[k].double() if inp[k].is_floating_point() else inp[k] for k in pkg.synth.FORWARD_ORDER]
                    l64 = O.forward_train(W, opt, *a64)
                    (l64[0] + w['w_att2'] * l64[1] + w['w_grd'] * l64[2] + w['w_cls'] * l64[3]).backward()
                finally:
                    torch.set_default_dtype(torch.float32)
                out.update(grad_norms_f64=np.array([float(W[n].grad.norm()) for n in names], dtype=np.float64),
                           grad_proj_f64=np.array([cases.grad_projections(n, W[n].grad.float()) for n in names], dtype=np.float64),
                           losses_f64=np.array([float(x) for x in l64[:4]], dtype=np.float64))
            if spec.get('bn_train'):     # the batch-statistics pass also moved the running statistics (momentum 0.1)
                bn = dict(ref.named_buffers())
                out.update(bn_running_mean=bn['att_embed_aux.0.running_mean'].numpy().copy(),
                           bn_running_var=bn['att_embed_aux.0.running_var'].numpy().copy())
        else:
            with torch.no_grad():
                lm, a2, gl, cl = ref(*args, 'MLE')
        out['losses'] = np.array([lm.item(), a2.item(), gl.item(), cl.item()], dtype=np.float32)
    elif spec['mode'] == 'dp':
        # one replica per shard, exactly what nn.DataParallel scatters (main.py:654-655): per-shard masked-mean losses,
        # total loss = sum over replicas / n_replicas (main.py:238-255), gradients accumulate over the replicas
        w = cases.GRAD_WEIGHTS
        n = spec['shards']
        per = spec['B'] // n
        ref.zero_grad()
        shard_losses = []
        for r in range(n):
            sub = {k: v[r * per:(r + 1) * per].contiguous() for k, v in inp.items()}     # scatter of the trimmed batch
            lm, a2, gl, cl = ref(*pkg.synth.as_args(sub), 'MLE')
            ((lm.sum() + w['w_att2'] * a2.sum() + w['w_grd'] * gl.sum() + w['w_cls'] * cl.sum()) / n).backward()
            shard_losses.append([lm.item(), a2.item(), gl.item(), cl.item()])
            print('   shard %d/%d  %.1fs  losses %s' % (r + 1, n, time.time() - t0, shard_losses[-1]), flush=True)
        names, norms, projs = [], [], []
        for pn, p in ref.named_parameters():
            if p.grad is not None:
                names.append(pn)
                norms.append(float(p.grad.double().norm()))
                projs.append(cases.grad_projections(pn, p.grad))
        out.update(shard_losses=np.array(shard_losses, dtype=np.float32), grad_names=np.array(names),
                   grad_norms=np.array(norms, dtype=np.float64), grad_proj=np.array(projs, dtype=np.float64),
                   losses=np.array(shard_losses, dtype=np.float64).mean(0).astype(np.float32))
    elif spec['mode'] == 'step':
        # main.train (main.py:234-266) with eval-mode arithmetic (dropout off, BN running stats: the only mode in which
        # CPU and GPU runs are comparable); optimizer exactly as main.py:660-677 builds it
        w = cases.GRAD_WEIGHTS
        lr, clip = 5e-4, 0.1
        params = []
        for key, value in dict(ref.named_parameters()).items():
            if value.requires_grad:
                params += [{'params': [value], 'lr': lr * (0.1 if ('ctx2pool_grd' in key or 'vis_embed' in key) else 1.0),
                            'weight_decay': 0, 'betas': (0.9, 0.999)}]
        optimizer = torch.optim.Adam(params)
        before = {n: p.detach().clone() for n, p in ref.named_parameters()}
        lm, a2, gl, cl = ref(*args, 'MLE')
        loss = (lm.sum() + w['w_att2'] * a2.sum() + w['w_grd'] * gl.sum() + w['w_cls'] * cl.sum()) / lm.numel()
        ref.zero_grad()
        loss.backward()
        total = torch.nn.utils.clip_grad_norm_(ref.parameters(), clip)
        optimizer.step()
        names, dn, mn, mp, dp = [], [], [], [], []
        for n, p in ref.named_parameters():
            if p.grad is None:
                continue
            names.append(n)
            dn.append(float((p.detach() - before[n]).double().norm()))
            mn.append(float(optimizer.state[p]['exp_avg'].double().norm()))
            # direction pins: the first moment (linear in the clipped gradient) and the parameter update itself
            mp.append(cases.grad_projections(n, optimizer.state[p]['exp_avg']))
            dp.append(cases.grad_projections(n, p.detach() - before[n]))
        out.update(step_names=np.array(names), delta_norms=np.array(dn), exp_avg_norms=np.array(mn),
                   exp_avg_proj=np.array(mp, dtype=np.float64), delta_proj=np.array(dp, dtype=np.float64),
                   total_grad_norm=np.float64(float(total)), loss=np.float64(float(loss)),
                   losses=np.array([lm.item(), a2.item(), gl.item(), cl.item()], dtype=np.float32))
    elif spec['mode'] == 'traj':
        # `steps` consecutive steps of main.train (main.py:234-266, optimizer of 660-677), eval-mode arithmetic, one batch each
        w = cases.GRAD_WEIGHTS
        lr, clip = 5e-4, 0.1
        params = []
        for key, value in dict(ref.named_parameters()).items():
            if value.requires_grad:
                params += [{'params': [value], 'lr': lr * (0.1 if ('ctx2pool_grd' in key or 'vis_embed' in key) else 1.0),
                            'weight_decay': 0, 'betas': (0.9, 0.999)}]
        optimizer = torch.optim.Adam(params)
        before = {n: p.detach().clone() for n, p in ref.named_parameters()}
        losses, norms, step_dn = [], [], []
        for batch in cases.traj_batches(name):
            lm, a2, gl, cl = ref(*pkg.synth.as_args(batch), 'MLE')
            loss = (lm.sum() + w['w_att2'] * a2.sum() + w['w_grd'] * gl.sum() + w['w_cls'] * cl.sum()) / lm.numel()
            ref.zero_grad()
            loss.backward()
            norms.append(float(torch.nn.utils.clip_grad_norm_(ref.parameters(), clip)))
            prev = {n: p.detach().clone() for n, p in ref.named_parameters() if p.grad is not None}
            optimizer.step()
            # norm of THIS step's whole update (all parameters): the bias-corrected step size of every step, not only their sum
            step_dn.append(float(torch.sqrt(sum(((p.detach() - prev[n]).double() ** 2).sum() for n, p in ref.named_parameters()
                                                 if n in prev))))
            losses.append([lm.item(), a2.item(), gl.item(), cl.item()])
            print('   step %d  %.1fs  losses %s  |grad| %.5f' % (len(losses), time.time() - t0, losses[-1], norms[-1]), flush=True)
        names, dn, dp = [], [], []
        mn, mp, vn, vp, st = [], [], [], [], []
        for n, p in ref.named_parameters():
            if p.grad is None:
                continue
            names.append(n)
            dn.append(float((p.detach() - before[n]).double().norm()))
            dp.append(cases.grad_projections(n, p.detach() - before[n]))
            # the optimiser state after the last step (main.py:660-677's torch.optim.Adam): step count, first / second moment
            s_ = optimizer.state[p]
            st.append(float(s_['step']))
            mn.append(float(s_['exp_avg'].double().norm())); mp.append(cases.grad_projections(n, s_['exp_avg']))
            vn.append(float(s_['exp_avg_sq'].double().norm())); vp.append(cases.grad_projections(n, s_['exp_avg_sq']))
        out.update(step_losses=np.array(losses, dtype=np.float32), step_grad_norms=np.array(norms, dtype=np.float64),
                   step_names=np.array(names), delta_norms=np.array(dn), delta_proj=np.array(dp, dtype=np.float64),
                   step_delta_norms=np.array(step_dn, dtype=np.float64),
                   state_steps=np.array(st, dtype=np.float64), exp_avg_norms=np.array(mn, dtype=np.float64),
                   exp_avg_proj=np.array(mp, dtype=np.float64), exp_avg_sq_norms=np.array(vn, dtype=np.float64),
                   exp_avg_sq_proj=np.array(vp, dtype=np.float64),
                   losses=np.array(losses[0], dtype=np.float32))
    elif spec['mode'] == 'GRD':
        with torch.no_grad():
            cp, ai, gi = ref(*args, 'GRD')
        out.update(cls_pred=cp.numpy(), att2_ind=ai.numpy().astype(np.int16), grd_ind=gi.numpy().astype(np.int16))
    else:
        raise ValueError(spec['mode'])
    out['ref_seconds'] = np.float64(time.time() - t0)
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    np.savez_compressed(os.path.join(GOLDEN_DIR, name + '.npz'), **out)
    print('%-40s %6.1fs  %s' % (name, time.time() - t0,
                                {k: (v.shape if hasattr(v, 'shape') else v) for k, v in out.items()
                                 if k in ('seq', 'losses', 'cls_pred')}))


if __name__ == '__main__':
    torch.set_num_threads(os.cpu_count())
    todo = sys.argv[1:] or list(cases.CASES)
    for n in todo:
        run_case(n)
