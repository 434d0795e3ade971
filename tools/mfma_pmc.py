"""Workload for MFMA-utilisation PMC passes: the hot-path GEMM shapes on the pipelined kernel (fc7, fused QKV), the
decode-step shapes north_star names (attn_hid = both h2att queries, logit) and the padded-head flash attention kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gvd_amd
from gvd_amd import ops
def gemm(M, N, K, n=3):
    A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
    out = torch.empty(M, N, device='cuda')
    for _ in range(n):
        ops.gemm_nt(A, W, b, 0, out=out)
gemm(256000, 2048, 2048)        # fc7                      -> gemm_pipe_kernel<true>, grid 32000
gemm(256000, 3168, 1024)        # fused q|k|v              -> grid 50000
gemm(256, 5000, 1024, 6)        # logit   (decode step)    -> gemm_nt_kernel<64,64>
gemm(256, 1024, 1024, 6)        # attn_hid (both h2att queries stacked)
qkv = torch.randn(256, 1000, 18 * ops.HEAD_PAD, device='cuda') * 0.5
qkv.view(256, 1000, 18, ops.HEAD_PAD)[..., 171:] = 0
for _ in range(3):
    ops.flash_attn_padded(qkv, 6, 1.0 / 32.0)
torch.cuda.synchronize()
