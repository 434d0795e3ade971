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
    def ref():
        hs = []
        for qh, kh, vh in zip(q.chunk(6, -1), k.chunk(6, -1), v.chunk(6, -1)):
            hs.append(torch.matmul(torch.softmax(torch.matmul(qh, kh.transpose(1, 2)), -1), vh))
        return torch.cat(hs, -1)
    a = timeit(lambda: ops.flash_attn_heads(q, k, v, sizes)); b = timeit(ref)
    fl = B * 6 * 2 * 2 * 1000 * 1000 * 171
    print('B=%d: flash %.2f ms (%.1f TF/s useful) | bmm+softmax+bmm+cat %.2f ms | maxdiff %.2e' % (B, a, fl / a / 1e9, b, (ops.flash_attn_heads(q, k, v, sizes) - ref()).abs().max().item()))
