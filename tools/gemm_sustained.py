"""Sustained-load check of the pipelined fp32-MFMA GEMM: the same product back to back for several seconds, TF/s per
window — separates the burst rate the micro-benchmark sees (cool chip, boost clock) from the rate under the continuous
fp32-MFMA load of the B=256 preamble (power-limited clock).

    python tools/gemm_sustained.py [M N K [seconds]]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gvd_amd  # noqa: F401
from gvd_amd import ops

M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (256000, 2048, 2048)
secs = float(sys.argv[4]) if len(sys.argv) >= 5 else 5.0
# variants of the in-situ call: GS_MDEV=<live fraction> (device-side row count), GS_ROWMAP=1 (row gather fused into the
# operand loads, live rows scattered over a 1.25x larger source), GS_RELU=1 (post-ReLU operand: half zeros)
mfrac = float(os.environ.get('GS_MDEV', '0'))
A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
if os.environ.get('GS_RELU') == '1':
    A.relu_()
out = torch.empty(M, N, device='cuda')
kw = {}
live = M
if mfrac > 0:
    live = int(M * mfrac)
    kw['m_dev'] = torch.tensor([live], dtype=torch.int32, device='cuda')
if os.environ.get('GS_ROWMAP') == '1':
    src = torch.randn(int(M * 1.25), K, device='cuda')
    keep = torch.rand(src.shape[0], device='cuda').argsort()[:M].sort()[0].to(torch.int32).contiguous()
    A, kw['a_row_map'] = src, keep
fl = 2.0 * live * N * K
_gemm = ops.gemm_nt
ops_gemm = lambda: _gemm(A, W, b, 1, out=out, **kw)
for _ in range(2):
    ops_gemm()
torch.cuda.synchronize()
win = 10
t_end = time.perf_counter() + secs
rates = []
while time.perf_counter() < t_end:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(win):
        ops_gemm()
    e1.record()
    e1.synchronize()
    rates.append(fl * win / (e0.elapsed_time(e1) * 1e-3) / 1e12)
print('[mdev=%s rowmap=%s relu=%s]' % (os.environ.get('GS_MDEV'), os.environ.get('GS_ROWMAP'), os.environ.get('GS_RELU')), end=' ')
print('M=%d N=%d K=%d: %d windows of %d launches; TF/s first %.1f, max %.1f, last %.1f, mean of the last half %.1f'
      % (M, N, K, len(rates), win, rates[0], max(rates), rates[-1], sum(rates[len(rates) // 2:]) / max(1, len(rates) - len(rates) // 2)))
print('trace:', ' '.join('%.0f' % r for r in rates[::max(1, len(rates) // 30)]))
