"""GEMM micro-benchmark: gvd gemm_nt (MFMA fp32) vs torch (rocBLAS) on the hot-path shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gvd_amd
from gvd_amd import ops
def timeit(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for M, N, K, act in ((256000, 2048, 2048, 1), (256000, 512, 1024, 0), (256000, 433, 2048, 0), (64000, 2048, 2048, 1)):
    A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
    out = torch.empty(M, N, device='cuda')
    t1 = timeit(lambda: ops.gemm_nt(A, W, b, act, out=out))
    t2 = timeit(lambda: torch.relu_(torch.addmm(b, A, W.t())) if act else torch.addmm(b, A, W.t()))
    fl = 2.0 * M * N * K
    print('M=%d N=%d K=%d: gvd %.2f ms (%.1f TF/s) | rocBLAS %.2f ms (%.1f TF/s)' % (M, N, K, t1 * 1e3, fl / t1 / 1e12, t2 * 1e3, fl / t2 / 1e12))
