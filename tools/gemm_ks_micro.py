"""Micro-benchmark of the token-loop products of a SMALL batch (LSTM cell with its fused epilogue, attention queries, logits)
against fp64, through ops.lstm_cell / ops.gemm_nt - i.e. whatever kernel the library routes the shape to.
    python tools/gemm_ks_micro.py [B ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gvd_amd  # noqa: E402,F401
from gvd_amd import ops  # noqa: E402

H, E, V = 1024, 512, 5000


def timed(f, n=40, warm=5):
    for _ in range(warm):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in ([int(a) for a in sys.argv[1:]] or [24, 32, 64, 96]):
    g = torch.Generator().manual_seed(B)
    w_ih, w_hh = torch.randn(4 * H, H + E, generator=g).cuda() / 32, torch.randn(4 * H, H, generator=g).cuda() / 32
    b_ih, b_hh = torch.randn(4 * H, generator=g).cuda() * 0.1, torch.randn(4 * H, generator=g).cuda() * 0.1
    x1, x2 = torch.randn(B, H, generator=g).cuda(), torch.randn(B, E, generator=g).cuda()
    h, c = torch.randn(B, H, generator=g).cuda() * 0.5, torch.randn(B, H, generator=g).cuda()
    gates = torch.empty(B, 4 * H, device='cuda')
    f = lambda: ops.lstm_cell([x1, x2], [w_ih[:, :H], w_ih[:, H:]], h, w_hh, b_ih, b_hh, c, gates_out=gates)
    oh, oc = f()
    G = (torch.cat([x1, x2], 1).double() @ w_ih.double().t() + h.double() @ w_hh.double().t() + b_ih.double() + b_hh.double())
    i, ff, gg, o = G.chunk(4, 1)
    rc = torch.sigmoid(ff) * c.double() + torch.sigmoid(i) * torch.tanh(gg)
    rh = torch.sigmoid(o) * torch.tanh(rc)
    us = timed(f)
    print('B=%3d LSTM cell (K=%d): %6.1f us  %6.1f TF/s  |h err| %.2e |c err| %.2e' %
          (B, 2 * H + E, us, 2.0 * B * 4 * H * (2 * H + E) / us / 1e6, float((oh.double() - rh).abs().max()), float((oc.double() - rc).abs().max())), flush=True)
    ws, bs = torch.randn(1024, H, generator=g).cuda() / 32, torch.randn(1024, generator=g).cuda() * 0.1
    f2 = lambda: ops.gemm_nt(h, ws, bs)
    q = f2()
    us = timed(f2)
    print('B=%3d queries  [%d x 1024] K=1024: %6.1f us  |err| %.2e' % (B, B, us, float((q.double() - (h.double() @ ws.double().t() + bs.double())).abs().max())), flush=True)
    wl, bl = torch.randn(V, H, generator=g).cuda() / 32, torch.randn(V, generator=g).cuda() * 0.1
    f3 = lambda: ops.gemm_nt(h, wl, bl)
    lg = f3()
    us = timed(f3)
    print('B=%3d logits   [%d x %d] K=1024: %6.1f us  |err| %.2e' % (B, B, V, us, float((lg.double() - (h.double() @ wl.double().t() + bl.double())).abs().max())), flush=True)
