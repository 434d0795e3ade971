"""Timing of the HBM-bound row kernels at the B=256 shapes.  python tools/rowwise_micro.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gvd_amd
from gvd_amd import ops
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
B, R = 256, 1000
x = torch.randn(B, R, 1024, device='cuda'); y = torch.randn(B, R, 1024, device='cuda')
g = torch.ones(1024, device='cuda'); b = torch.zeros(1024, device='cuda')
t = timeit(lambda: ops.add_layernorm_unbiased(x, y, g, b))
print('add_layernorm_unbiased [256000,1024]: %.3f ms  (%.2f TB/s of 3.15 GB)' % (t, 3 * x.numel() * 4 / t / 1e9))
gp = torch.relu(torch.randn(B, R, 2048, device='cuda')); loc = torch.randn(B, R, 300, device='cuda')
lg = torch.randn(B, R, 433, device='cuda'); pm = (torch.rand(B, R + 1, device='cuda') < 0.2).to(torch.uint8)
t = timeit(lambda: ops.region_feature_rows(gp, loc, lg, pm))
nbytes = B * R * (2048 + 300 + 433 + 2781 + 433) * 4
print('region_feature_rows [256000 rows]: %.3f ms  (%.2f TB/s of %.2f GB)' % (t, nbytes / t / 1e9, nbytes / 1e9))
