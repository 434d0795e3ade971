"""GRU micro-benchmark: persistent HIP kernel vs torch nn.GRU (MIOpen).  python tools/gru_micro.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gvd_amd
from gvd_amd import ops
torch.manual_seed(0)
gru = torch.nn.GRU(1024, 512, 2, dropout=0.2, bidirectional=True, batch_first=True).cuda().eval()
def timeit(f, n):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for B, T, n in ((4, 480, 5), (32, 480, 5), (160, 480, 3), (256, 480, 3), (256, 10, 10)):
    x = torch.randn(B, T, 1024, generator=torch.Generator().manual_seed(B * T)).cuda()
    with torch.no_grad():
        a = timeit(lambda: ops.gru_bidir_2layer(x, gru, barrier='counter'), n)
        c = timeit(lambda: ops.gru_bidir_2layer(x, gru, barrier='cg'), n)
        b = timeit(lambda: gru(x)[0], n)
        d = (ops.gru_bidir_2layer(x, gru, barrier='counter') - gru(x)[0]).abs().max().item()
        bits = int(ops.gru_bidir_2layer(x, gru, barrier='counter').view(torch.int32).to(torch.int64).sum())
    print('B=%d T=%d: hip counter-barrier %.2f ms (%.1f us/step/layer), cg-sync %.2f ms, miopen %.2f ms, maxdiff %.2e, output bits sum %d'
          % (B, T, a, a * 1e3 / (2 * T), c, b, d, bits))
