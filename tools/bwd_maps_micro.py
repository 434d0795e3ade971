"""Time the backward maps kernel of the training attention core alone (B=64, Rp=1024, p=0.2) - used with tools/with_cflags.py
to time ablated builds (-DGVD_BWD_ABL=n)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gvd_amd  # noqa: E402,F401
from gvd_amd import ops  # noqa: E402
from gvd_amd.hip import check, lib, ptr, stream_ptr  # noqa: E402
Bt, R, Rp, nh, HP = 64, 1000, 1024, 6, ops.HEAD_PAD
W3 = 3 * nh * HP
dev = 'cuda'
qkv = torch.zeros(Bt, Rp, 3, nh, HP, device=dev)
qkv[..., :171] = torch.randn(Bt, Rp, 3, nh, 171, device=dev) * 0.5
qkv = qkv.reshape(Bt, Rp, W3)
dO = torch.randn(Bt, Rp, nh * HP, device=dev)
O = torch.randn(Bt, Rp, nh * HP, device=dev)
lse = torch.zeros(Bt * nh, Rp, device=dev)
delta = torch.empty(Bt * nh, Rp, device=dev)
Pd = torch.empty(Bt, nh, Rp, Rp, device=dev)
dS = torch.empty(Bt, nh, Rp, Rp, device=dev)
f = lambda: check(lib().gvd_enc_attn_bwd_maps(ptr(qkv), W3, ptr(dO), ptr(O), nh * HP, ptr(lse), None, ptr(delta), ptr(Pd), ptr(dS),
                                              Bt, Rp, R, nh, HP, 1.0 / 32, 0.2, 12345, stream_ptr()), 'maps')
for _ in range(2):
    f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(5):
    f()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print('bwd maps (delta + maps kernel) B=%d: %.3f ms  %.1f TF/s' % (Bt, ms, 2 * Bt * nh * 2.0 * R * R * HP / ms / 1e9), flush=True)
