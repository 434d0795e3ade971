"""Micro-benchmark of the streaming products / the small-M dX launch / the row kernels at the training-step shapes
(B = 64 segments, Lc = 20 words, R = 1000 regions): HIP-event time per launch, algorithmic bytes -> GB/s, and the torch
(rocBLAS / ATen) formulation each one replaced, timed the same way.  One JSON line per kernel.

    python tools/stream_mm_bench.py [B] [reps]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gvd_amd  # noqa: E402,F401
from gvd_amd import ops  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    only = sys.argv[3] if len(sys.argv) > 3 else ''          # substring filter on the kernel names
    M, R, K, H = 20, 1000, 2048, 1024
    g = torch.Generator().manual_seed(0)
    dev = 'cuda'
    xt = torch.randn(B, M, K, generator=g).to(dev)
    feats = torch.randn(B, R, K, generator=g).to(dev)
    mask = (torch.rand(B, M, R, generator=g) < 0.3).to(torch.uint8).to(dev)
    mb, rb = torch.randn(B, M, generator=g).to(dev), torch.randn(B, M, R, generator=g).to(dev)
    dout = torch.randn(B, M, R, generator=g).to(dev)
    dout_t = ops.masked_copy_rowsum(dout, None, want_sum=False, want_t=True)[2]
    out = []

    def rec(name, fn, nbytes, base_fn=None, base=None):
        if only and only not in name:
            return
        us = timed(fn, reps)
        base_us = timed(base_fn, reps) if (base_fn is not None and not only) else None
        d = {'kernel': name, 'us': round(us, 2), 'GBs': round(nbytes / us / 1e3, 1), 'frac_hbm_8TBs': round(nbytes / us / 1e3 / 8000, 4),
             'bytes': nbytes}
        if base_us is not None:
            d.update(replaced=base, replaced_us=round(base_us, 2))
        if os.environ.get('GVD_GS_VARIANT'):
            d['GVD_GS_VARIANT'] = os.environ['GVD_GS_VARIANT']
        print(json.dumps(d), flush=True)
        out.append(d)

    nb_f = 4 * (B * R * K + B * M * K + 2 * B * M * R + B * M) + B * M * R
    rec('grounder_fwd (gvd_grounder_fwd_f32)', lambda: ops.grounder_stream(xt, feats, mask, mb, rb), nb_f,
        lambda: ops.grounder_dot(xt, feats, mask, mb, rb), 'gemm_nt_kernel<32,128> batched (round 4)')
    rec('rows_contract gather form', lambda: ops.rows_contract(dout, feats), 4 * (B * R * K + B * M * R + B * M * K))
    rec('rows_contract d xt (gvd_rows_contract_f32, transposed S)', lambda: ops.rows_contract(dout, feats, S_t=dout_t),
        4 * (B * R * K + B * M * R + B * M * K), lambda: torch.matmul(dout, feats), 'torch.matmul (rocBLAS bmm)')
    rec('rank_update d feats N=2048 (gvd_rank_update_f32)', lambda: ops.rank_update(dout, xt),
        4 * (B * R * K + B * M * R + B * M * K), lambda: torch.matmul(dout.transpose(1, 2), xt),
        'torch.matmul (rocBLAS bmm)')
    dctx = torch.randn(M, B, 2 * H, generator=g).to(dev)[:, :, :H].transpose(0, 1)
    alpha = torch.softmax(torch.randn(B, M, R, generator=g), -1).to(dev)
    rec('rank_update d pool N=1024', lambda: ops.rank_update(alpha, dctx), 4 * (B * R * H + B * M * R + B * M * H),
        lambda: torch.bmm(alpha.transpose(1, 2), dctx), 'torch.bmm')
    # the BPTT step's products
    dg_l, dg_a = torch.randn(B, 4 * H, generator=g).to(dev), torch.randn(B, 4 * H, generator=g).to(dev)
    w_ih, w_hh, w_ahh = (torch.randn(4 * H, n, generator=g).to(dev) * 0.05 for n in (2 * H, H, H))
    dX, dh, dha = torch.empty(B, 2 * H, device=dev), torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)
    groups = [dict(A=dg_l, W=w_ih, out=dX), dict(A=dg_l, W=w_hh, out=dh), dict(A=dg_a, W=w_ahh, out=dha)]

    def lib3():
        torch.mm(dg_l, w_ih, out=dX); torch.mm(dg_l, w_hh, out=dh); torch.mm(dg_a, w_ahh, out=dha)
    rec('dx_products 3 groups (gvd_gemm_dx_small_f32), M=%d' % B, lambda: ops.dx_products(groups, B),
        4 * 4 * H * 4 * H, lib3, '3 x torch.mm (rocBLAS)')
    dq, wst = torch.randn(B, 1024, generator=g).to(dev), torch.randn(1024, H, generator=g).to(dev)
    o2 = torch.empty(B, H, device=dev)
    rec('dx_products addend form (d h_att)', lambda: ops.dx_products([dict(A=dq, W=wst, out=o2, addend=dX[:, H:])], B),
        4 * 1024 * H, lambda: torch.addmm(dX[:, H:], dq, wst, out=o2), 'torch.addmm')
    scores = torch.randn(B, M, R, generator=g).to(dev)
    rec('softmax_rows [B,Lc,R]', lambda: ops.softmax_rows(scores), 8 * B * M * R,
        lambda: torch.softmax(scores, -1), 'torch.softmax')
    x = torch.randn(B * 10, 1024, generator=g).to(dev).requires_grad_(True)
    bn = torch.nn.BatchNorm1d(1024).to(dev).train()
    rec('bn_relu_train fwd [B*10,1024]', lambda: ops.bn_relu_train(x.detach(), bn), 8 * x.numel(),
        lambda: torch.relu(bn(x.detach().view(B, 10, 1024).permute(0, 2, 1).contiguous())).permute(0, 2, 1).contiguous(),
        'permute + nn.BatchNorm1d + relu + permute')


if __name__ == '__main__':
    main()
