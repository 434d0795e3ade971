"""Attention micro-benchmark: first flash kernel (uneven heads in place) vs the padded-head kernel vs the library chain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gvd_amd
from gvd_amd import ops
def timeit(f, n=3):
    for _ in range(1): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for B in (4, 64, 256):
    q = torch.randn(B, 1000, 1024, device='cuda') * 0.3; k = torch.randn(B, 1000, 1024, device='cuda'); v = torch.randn(B, 1000, 1024, device='cuda')
    sizes = [t.shape[-1] for t in q[:1, :1].chunk(6, -1)]
    HP = ops.HEAD_PAD
    qkv = torch.zeros(B, 1000, 18 * HP, device='cuda')
    c0 = 0
    for h, w in enumerate(sizes):
        for j, t in enumerate((q, k, v)):
            qkv[:, :, (j * 6 + h) * HP:(j * 6 + h) * HP + w] = t[:, :, c0:c0 + w]
        c0 += w
    def ref():
        hs = []
        for qh, kh, vh in zip(q.chunk(6, -1), k.chunk(6, -1), v.chunk(6, -1)):
            hs.append(torch.matmul(torch.softmax(torch.matmul(qh, kh.transpose(1, 2)), -1), vh))
        return torch.cat(hs, -1)
    a = timeit(lambda: ops.flash_attn_heads(q, k, v, sizes)); c = timeit(lambda: ops.flash_attn_padded(qkv, 6, 1.0))
    b = timeit(ref) if B <= 64 else float('nan')
    fl = B * 6 * 2 * 2 * 1000 * 1000 * 171
    o = ops.flash_attn_padded(qkv, 6, 1.0); o1 = ops.flash_attn_heads(q, k, v, sizes)
    d = max((o[:, :, h * HP:h * HP + w] - o1[:, :, sum(sizes[:h]):sum(sizes[:h]) + w]).abs().max().item() for h, w in enumerate(sizes))
    print('B=%d: flash16 %.2f ms (%.1f TF/s useful) | padded %.2f ms (%.1f TF/s useful, %.1f incl. pads) | bmm+softmax+bmm+cat %.2f ms | maxdiff %.2e'
          % (B, a, fl / a / 1e9, c, fl / c / 1e9, fl / c / 1e9 * 176 / 171, b, d), flush=True)
