"""One pipelined-GEMM shape at a time, HIP-event timed: forward (A[M,K] W[N,K]^T), dX (dY[M,N] W[N,K] -> [M,K]) or dW
(dY[M,N]^T X[M,K] -> [N,K]); prints microseconds and TFLOP/s of each `form:M,N,K` argument.
    python tools/gemm_shape_micro.py dx:64000,1024,1056 dw:64000,1024,1056 fwd:64000,448,2048"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gvd_amd  # noqa: E402,F401
from gvd_amd import ops  # noqa: E402


def timed(f, n=10, warm=3):
    for _ in range(warm):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for arg in sys.argv[1:]:
    form, dims = arg.split(':')
    M, N, K = (int(x) for x in dims.split(','))
    if form == 'fwd':
        a, w = torch.randn(M, K, device='cuda'), torch.randn(N, K, device='cuda')
        f = lambda: ops.gemm_nt(a, w)
    elif form == 'dx':
        dy, w = torch.randn(M, N, device='cuda'), torch.randn(N, K, device='cuda')
        assert ops.gemm_dx(dy, w) is not None
        f = lambda: ops.gemm_dx(dy, w)
    elif form == 'dxa':
        dy, w, add = torch.randn(M, N, device='cuda'), torch.randn(N, K, device='cuda'), torch.randn(M, K, device='cuda')
        f = lambda: ops.gemm_dx(dy, w, addend=add)
    else:
        dy, x = torch.randn(M, N, device='cuda'), torch.randn(M, K, device='cuda')
        assert ops.gemm_dw(dy, x) is not None
        f = lambda: ops.gemm_dw(dy, x)
    ms = timed(f)
    print('%-4s M=%d N=%d K=%d: %.1f us  %.1f TF/s (%.3f of 157.3)  GVD_GEMM_EDGE=%s'
          % (form, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9, 2.0 * M * N * K / ms / 1e9 / 157.3,
             os.environ.get('GVD_GEMM_EDGE', '0')), flush=True)
