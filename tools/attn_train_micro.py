"""Micro-benchmark of the encoder's attention core: inference flash kernel (dense B=256 / ragged like the compacted preamble)
and the TRAINING core at the configs[2] shape (B=64, R=1000 -> Rp=1024, 6 heads x 176 columns): flash-style forward with
dropout, backward maps kernel, the three N=176 K-strided products.  HIP-event timings per stage.
    python tools/attn_train_micro.py [B_train] [B_infer]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gvd_amd  # noqa: E402,F401
from gvd_amd import ops  # noqa: E402
from gvd_amd.hip import check, lib, ptr, stream_ptr  # noqa: E402

Bt = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Bi = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = 'cuda'
nh, HP = 6, ops.HEAD_PAD


def timed(f, n=5, warm=2):
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


def packed(B, R):
    qkv = torch.zeros(B, R, 3, nh, HP, device=dev)
    qkv[..., :171] = torch.randn(B, R, 3, nh, 171, device=dev) * 0.5
    return qkv.reshape(B, R, 3 * nh * HP)


# ---- inference: dense and ragged (compacted-preamble shape: ~800 valid rows + 1 weighted representative per sample)
if Bi > 0:
    R = 1000
    qkv = packed(Bi, R)
    ms = timed(lambda: ops.flash_attn_padded(qkv, nh, 1.0 / 32.0))
    fl = Bi * nh * 4.0 * R * R * HP
    print('flash inference dense  B=%d R=%d: %.3f ms  %.1f TF/s incl. pads (%.3f of 157.3)' % (Bi, R, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3), flush=True)
    g = torch.Generator().manual_seed(0)
    nv = (800 + torch.randint(-25, 26, (Bi,), generator=g)).tolist()
    off = [0]
    for n in nv:
        off.append(off[-1] + n + 1)
    cap = Bi * (R + 1)
    qc = packed(1, cap).view(cap, -1)
    offd = torch.tensor(off, dtype=torch.int32, device=dev)
    kw = torch.log2(torch.tensor([float(R - n) for n in nv], device=dev))
    ms = timed(lambda: ops.flash_attn_padded(qc, nh, 1.0 / 32.0, ragged=(Bi, R + 1, offd, kw)))
    fl = sum(nh * 4.0 * (n + 1) * (n + 1) * HP for n in nv)
    print('flash inference ragged B=%d rows~801: %.3f ms  %.1f TF/s incl. pads (%.3f of 157.3)' % (Bi, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3), flush=True)
    # where does the ragged shape lose against the dense one?  the same row count for every sample, dense vs ragged
    for n in (767, 800):
        rows = n + 1
        qd = packed(Bi, rows)
        ms_d = timed(lambda: ops.flash_attn_padded(qd, nh, 1.0 / 32.0))
        offu = torch.arange(0, (Bi + 1) * rows, rows, dtype=torch.int32, device=dev)
        kwu = torch.log2(torch.full((Bi,), float(R - n), device=dev))
        qr = qd.view(Bi * rows, -1)
        ms_r = timed(lambda: ops.flash_attn_padded(qr, nh, 1.0 / 32.0, ragged=(Bi, rows, offu, kwu)))
        ms_r2 = timed(lambda: ops.flash_attn_padded(qr, nh, 1.0 / 32.0, ragged=(Bi, R + 1, offu, kwu)))
        fl = Bi * nh * 4.0 * rows * rows * HP
        print('flash %d rows per sample: dense %.3f ms (%.1f TF/s) | ragged, grid for %d rows %.3f ms | ragged, grid for %d rows %.3f ms'
              % (rows, ms_d, fl / ms_d / 1e9, rows, ms_r, R + 1, ms_r2), flush=True)
        del qd, qr
    del qkv, qc

# ---- training core
if Bt > 0:
    R, Rp = 1000, 1024
    W3 = 3 * nh * HP
    qkv = packed(Bt, Rp)
    dO = torch.randn(Bt, Rp, nh * HP, device=dev)
    dO[:, R:] = 0
    O = torch.zeros(Bt, Rp, nh * HP, device=dev)
    lse = torch.empty(Bt * nh, Rp, device=dev)
    delta = torch.empty(Bt * nh, Rp, device=dev)
    Pd = torch.empty(Bt, nh, Rp, Rp, device=dev)
    dS = torch.empty(Bt, nh, Rp, Rp, device=dev)
    dqkv = torch.zeros_like(qkv)
    Sc = torch.empty(Bt * nh, Rp, Rp, device=dev)      # the forward's scores, handed to the backward maps kernel
    prod = Bt * nh * 2.0 * R * R * HP          # one product's flops
    for p in (0.0, 0.2):
        fwd = lambda: check(lib().gvd_flash_attn_train_fwd_f32(ptr(qkv), W3, ptr(O), nh * HP, ptr(lse), ptr(Sc), Bt, Rp, R, Rp, nh, HP, 1.0 / 32,
                                                               None, p, 12345, stream_ptr()), 'fwd')
        ms = timed(fwd)
        print('train fwd  (flash, p=%.1f) B=%d: %.3f ms  %.1f TF/s (2 products)' % (p, Bt, ms, 2 * prod / ms / 1e9), flush=True)
        for use_s in (False, True):
            maps = lambda: check(lib().gvd_enc_attn_bwd_maps(ptr(qkv), W3, ptr(dO), ptr(O), nh * HP, ptr(lse), None, ptr(Sc) if use_s else None,
                                                             ptr(delta), ptr(Pd), ptr(dS), Bt, Rp, R, Rp, nh, HP, 1.0 / 32, p, 12345, stream_ptr()), 'maps')
            ms = timed(maps)
            print('train bwd maps (%s + epilogue, p=%.1f): %.3f ms, writes %.2f GB'
                  % ('scores loaded, dY' if use_s else 'S, dY', p, ms, 2 * Pd.numel() * 4 / 1e9), flush=True)
    ko, vo = nh * HP, 2 * nh * HP
    mb, msz = nh * Rp * Rp, Rp * Rp
    f_dv = lambda: ops._heads_bgemm(nh, Pd, 0, Rp, mb, msz, dO, 0, nh * HP, Rp * nh * HP, HP, Rp, dqkv, vo, W3, Rp * W3, HP, R, HP, Bt, a_t=1, w_t=1)
    f_dq = lambda: ops._heads_bgemm(nh, dS, 0, Rp, mb, msz, qkv, ko, W3, Rp * W3, HP, Rp, dqkv, 0, W3, Rp * W3, HP, R, HP, Bt, w_t=1)
    f_dk = lambda: ops._heads_bgemm(nh, dS, 0, Rp, mb, msz, qkv, 0, W3, Rp * W3, HP, Rp, dqkv, ko, W3, Rp * W3, HP, R, HP, Bt, a_t=1, w_t=1)
    for name, f in (('dV = Pd^T dO', f_dv), ('dQ = dS K', f_dq), ('dK = dS^T Q', f_dk)):
        ms = timed(f)
        print('train %-14s: %.3f ms  %.1f TF/s' % (name, ms, Bt * nh * 2.0 * R * Rp * HP / ms / 1e9), flush=True)
    q2 = qkv.clone().requires_grad_(True)

    def whole():
        o = ops.enc_attn_core(q2, R, nh, 1.0 / 32, 0.2, seed=7)
        o.backward(dO)
        q2.grad = None
    ms = timed(whole, n=3, warm=1)
    print('train core forward + backward (autograd Function, p=0.2): %.3f ms per layer' % ms, flush=True)
