"""Workload for MFMA-utilisation PMC passes: fc7-shaped GEMM (gvd + rocBLAS) and the flash attention kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gvd_amd
from gvd_amd import ops
M, N, K = 256000, 2048, 2048
A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
out = torch.empty(M, N, device='cuda')
for _ in range(3):
    ops.gemm_nt(A, W, b, 1, out=out)
    torch.addmm(b, A, W.t())
q = torch.randn(256, 1000, 1024, device='cuda') * 0.3; k = torch.randn(256, 1000, 1024, device='cuda'); v = torch.randn(256, 1000, 1024, device='cuda')
sizes = [t.shape[-1] for t in q[:1, :1].chunk(6, -1)]
for _ in range(3):
    ops.flash_attn_heads(q, k, v, sizes)
torch.cuda.synchronize()
