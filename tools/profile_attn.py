"""Run only the attention streaming kernel at the bench shape (for rocprofv3 --pmc passes):
B x (R regions + Ft temporal positions), A=512, H=1024.  python tools/profile_attn.py [B] [Ft] [iters] [R] [group]
(group = K > 1: B samples x K beam rows share each sample's features - the beam-search launch shape)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gvd_amd  # noqa: E402,F401
from gvd_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Ft = int(sys.argv[2]) if len(sys.argv) > 2 else 10
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
R = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
K = int(sys.argv[5]) if len(sys.argv) > 5 else 1
A, H = 512, 1024
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
pool = torch.randn(B, R, H, device=dev, generator=g)
p_pool = torch.randn(B, R, A, device=dev, generator=g)
conv = torch.randn(B, Ft, H, device=dev, generator=g)
p_conv = torch.randn(B, Ft, A, device=dev, generator=g)
q12 = torch.randn(B * K, 2 * A, device=dev, generator=g)
w1 = torch.randn(A, device=dev, generator=g) * 0.3
w2 = torch.randn(A, device=dev, generator=g) * 0.3
ab = torch.zeros(1, device=dev)
pm = (torch.rand(B, R + 1, device=dev, generator=g) < 0.2).to(torch.uint8).repeat_interleave(K, 0)
logits = torch.empty(B * K, R, device=dev)
region = dict(feats=pool, p_feats=p_pool, q=q12[:, A:], w=w2, alpha_bias=ab, att_mask=pm[:, 1:], pnt_mask=pm[:, 1:],
              logits_out=logits)
temporal = dict(feats=conv, p_feats=p_conv, q=q12[:, :A], w=w1, alpha_bias=ab)
if K > 1:
    region['group'] = K
    temporal['group'] = K
for _ in range(iters):
    out = ops.attention_step(region, temporal)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    out = ops.attention_step(region, temporal)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
nbytes = B * (R + Ft) * (A + H) * 4
# the streaming (partial) kernel alone: HIP event pairs around its launches
from gvd_amd import hip  # noqa: E402
timer = hip.KernelTimer(max_pairs=iters)
ops.set_kernel_timer(timer)
timer.reset()
for _ in range(iters):
    out = ops.attention_step(region, temporal)
torch.cuda.synchronize()
kms, kn = timer.read()
ops.set_kernel_timer(None)
print('partial kernel alone: %.1f us (%d launches), %.1f GB/s algorithmic' % (1e3 * kms / kn, kn, nbytes / (kms / kn) / 1e6))
print('attention_step (partial+combine) chunk=%s B=%d Ft=%d R=%d group=%d: %.1f us/call, %.1f GB/s algorithmic'
      % ('50', B, Ft, R, K, ms * 1e3, nbytes / ms / 1e6))
