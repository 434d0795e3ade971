"""Does a big fp32-MFMA GEMM run slower right after a phase of light (HBM-bound) kernels — i.e. is the in-situ rate of
the preamble GEMMs (124 TF/s for fc7 at the start of a step, 141 TF/s back to back) a clock / power-state ramp?
Per round: `light_ms` of streaming copies, then the fc7-shaped product 3 times, each timed with its own event pair."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gvd_amd  # noqa: F401
from gvd_amd import ops

M, N, K = 205000, 2048, 2048
A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
out = torch.empty(M, N, device='cuda')
x = torch.randn(64 << 20, device='cuda'); y = torch.empty_like(x)
fl = 2.0 * M * N * K
for _ in range(3):
    ops.gemm_nt(A, W, b, 1, out=out)
torch.cuda.synchronize()
for light_ms in (0, 2, 5, 15, 40):
    res = []
    for rnd in range(6):
        if light_ms:
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e3 < light_ms:
                y.copy_(x)                      # 512 MB streaming copy, ~0.1 ms each
                torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for i in range(3):
            ev[i].record()
            ops.gemm_nt(A, W, b, 1, out=out)
        ev[3].record()
        ev[3].synchronize()
        res.append([fl / (ev[i].elapsed_time(ev[i + 1]) * 1e-3) / 1e12 for i in range(3)])
    res = res[1:]
    print('light phase %2d ms -> TF/s of the 1st / 2nd / 3rd GEMM after it: %s' % (
        light_ms, ' / '.join('%.1f' % (sum(r[i] for r in res) / len(res)) for i in range(3))), flush=True)
# idle (no kernels at all) instead of light work
for idle_ms in (2, 15):
    res = []
    for rnd in range(6):
        torch.cuda.synchronize(); time.sleep(idle_ms * 1e-3)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for i in range(3):
            ev[i].record()
            ops.gemm_nt(A, W, b, 1, out=out)
        ev[3].record(); ev[3].synchronize()
        res.append([fl / (ev[i].elapsed_time(ev[i + 1]) * 1e-3) / 1e12 for i in range(3)])
    res = res[1:]
    print('idle %2d ms -> TF/s 1st / 2nd / 3rd: %s' % (idle_ms, ' / '.join('%.1f' % (sum(r[i] for r in res) / len(res)) for i in range(3))), flush=True)
