"""GEMM micro-benchmark: gvd gemm_nt (MFMA fp32) vs torch (rocBLAS) on the hot-path shapes.

    python tools/gemm_micro.py              # runs itself once per GVD_GEMM_VARIANT (1 = general 128x128, 3 = pipelined)
"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = (  # M, N, K, act : B=256 per-segment projections
    (256000, 2048, 2048, 1),     # fc7
    (256000, 433, 2048, 0),      # class logits
    (256000, 1024, 2784, 1),     # pool_embed (K padded 2781 -> 2784)
    (256000, 3168, 1024, 0),     # fused q|k|v, heads padded to 176
    (256000, 1024, 1056, 0),     # wo over padded heads
    (256000, 512, 1024, 1),      # feed-forward 1 / ctx2pool
    (256000, 1024, 512, 0),      # feed-forward 2
    (4000, 2048, 2048, 1),       # B=4 preamble
)


def child():
    import torch
    import gvd_amd  # noqa: F401
    from gvd_amd import ops

    def timeit(f, n=4):
        for _ in range(2):
            f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
    v = os.environ.get('GVD_GEMM_VARIANT', 'default')
    for M, N, K, act in SHAPES:
        A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
        out = torch.empty(M, N, device='cuda')
        t1 = timeit(lambda: ops.gemm_nt(A, W, b, act, out=out))
        fl = 2.0 * M * N * K
        line = 'variant %s M=%d N=%d K=%d: gvd %.3f ms (%.1f TF/s)' % (v, M, N, K, t1 * 1e3, fl / t1 / 1e12)
        if os.environ.get('GEMM_MICRO_LIB') == '1':
            t2 = timeit(lambda: torch.relu_(torch.addmm(b, A, W.t())) if act else torch.addmm(b, A, W.t()))
            line += ' | rocBLAS %.3f ms (%.1f TF/s)' % (t2 * 1e3, fl / t2 / 1e12)
        print(line, flush=True)
        del A, W, out


if __name__ == '__main__':
    if os.environ.get('GEMM_MICRO_CHILD'):
        child()
    else:
        for i, v in enumerate(sys.argv[1:] or ['1', '3']):
            env = dict(os.environ, GEMM_MICRO_CHILD='1', GVD_GEMM_VARIANT=v, GEMM_MICRO_LIB='1' if i == 0 else '0')
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
