"""Sweep of the split-K factor S of ops.gemm_dw (dW = dY^T X, both operands K-strided, S row chunks as one batched launch +
an ordered slab sum) over the weight-gradient shapes of the batch_size = 64 training step.
    python tools/dw_split_sweep.py            (one table line per shape and S: ms incl. the slab sum, useful TFLOP/s)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gvd_amd  # noqa: E402,F401
from gvd_amd import ops  # noqa: E402

# (round 6: the encoder layers run on the unpadded 64000 rows - S may be any divisor of M / 32 = 2000, not only a power of two)
SHAPES = [('fc7', 64000, 2048, 2048), ('pool_embed', 64000, 1024, 2784), ('sim_logits', 64000, 448, 2048),
          ('ctx2pool', 64000, 512, 1024), ('enc q|k|v', 64000, 3168, 1024), ('enc wo', 64000, 1024, 1056),
          ('enc ff1', 64000, 512, 1024), ('enc ff2', 64000, 1024, 512)]


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, M, N, K in SHAPES:
    dY = torch.randn(M, N, device='cuda')
    X = torch.randn(M, K, device='cuda')
    tiles = -(-N // 128) * -(-K // 128)
    ref = None
    cands = [s for s in (1, 2, 4, 5, 8, 10, 16, 20, 25, 40, 50, 80) if M % (32 * s) == 0 and tiles * s >= 256 and tiles * s <= 8192]
    default = ops.gemm_dw(dY, X)
    for S in cands:
        out = ops.gemm_dw(dY, X, split=S)
        if out is None:
            continue
        err = float((out - default).abs().max()) / float(default.abs().max())
        ms = timed(lambda: ops.gemm_dw(dY, X, split=S))
        print('%-11s M=%d N=%d K=%d tiles=%d S=%-2d wgs=%-5d %.3f ms  %.1f TF/s  rel.diff vs default %.1e'
              % (name, M, N, K, tiles, S, tiles * S, ms, 2.0 * M * N * K / ms / 1e9, err), flush=True)
    ms = timed(lambda: ops.gemm_dw(dY, X))
    print('%-11s default: %.3f ms  %.1f TF/s' % (name, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
    del dY, X
