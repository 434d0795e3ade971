import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gvd_amd
from gvd_amd import ops
from oracle import gvd_oracle as O
opt = gvd_amd.opts.default_opt(vocab_size=200)
inp = gvd_amd.synth.trim_to_batch(gvd_amd.synth.make_inputs(opt, 4, seed=1, train=True))
pm, fm = inp['pnt_mask'], inp['frm_mask']
ref = O.bbox_overlaps(inp['ppls'], inp['gt_boxes'], fm | pm[:, 1:].unsqueeze(-1))
ov, st = ops.iou_targets(inp['ppls'].cuda(), inp['gt_boxes'].cuda(), fm.cuda(), pm.cuda())
ov = ov.cpu()
d = (ov != ref)
print('mismatch', int(d.sum()), 'of', d.numel(), 'max abs', float((ov-ref).abs().max()))
idx = d.nonzero()[:8]
for b, r, k in idx.tolist():
    a = inp['ppls'][b, r, :5].numpy(); g = inp['gt_boxes'][b, k, :5].numpy()
    f32 = np.float32
    gx = f32(g[2]-g[0])+f32(1); gy = f32(g[3]-g[1])+f32(1); ax = f32(a[2]-a[0])+f32(1); ay = f32(a[3]-a[1])+f32(1)
    iw = f32(f32(min(a[2],g[2]) - max(a[0],g[0])) + f32(1)); ih = f32(f32(min(a[3],g[3]) - max(a[1],g[1])) + f32(1))
    inter = f32(iw*ih); ua = f32(f32(f32(ax*ay)+f32(gx*gy)) - inter)
    print(b, r, k, 'gpu %.9g ref %.9g numpy %.9g' % (ov[b,r,k], ref[b,r,k], f32(inter/ua)), 'iw,ih', iw, ih, 'ua', ua, 'masked', int(fm[b,r,k] | pm[b,1+r]))
