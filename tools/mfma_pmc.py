"""Workloads for MFMA-utilisation PMC passes, one section per invocation so that a pass's counters can be grouped by kernel
name alone:   python tools/mfma_pmc.py <section>
  fc7 | qkv            plain products of the preamble on the pipelined kernel (gemm_pipe_kernel, direct-to-LDS operands)
  dx | dw              K-strided backward products of nn.Linear (dX = dY W, dW = dY^T X), training shapes at B = 64
  attn_core            the training attention core: flash-style forward, the backward maps kernel (two products + fused
                       epilogue) and the three N = 176 K-strided products (dV, dQ, dK), B = 64, p_drop = 0.2
  ctx2pool             the fc7 -> attn_hid projection of the preamble (model.py:391: pool [B R, 1024] -> p_pool [B R, 512]), B = 256
  logit_train          the hidden -> |V| projection over the teacher-forced rows (B Lc = 1280 x 5000 x 1024), training
  logit | attn_hid | lstm   the per-token products, B = 256 rows (gemm_small_kernel): logit = hidden -> |V|, attn_hid = the two
                       stacked h2att queries (AttModel.py:39,77), lstm = one LSTM cell
  flash                padded-head flash attention (inference encoder), B = 256"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gvd_amd  # noqa: E402,F401
from gvd_amd import ops  # noqa: E402

sec = sys.argv[1]
dev = 'cuda'


def gemm(M, N, K, n=4):
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    for _ in range(n):
        ops.gemm_nt(A, W, b, 0, out=out)


if sec == 'fc7':
    gemm(256000, 2048, 2048)
elif sec == 'qkv':
    gemm(256000, 3168, 1024)
elif sec == 'ctx2pool':
    gemm(256000, 512, 1024)
elif sec == 'logit_train':
    gemm(1280, 5000, 1024, 8)
elif sec == 'dx':
    dY = torch.randn(64 * 1024, 3168, device=dev)
    W = torch.randn(3168, 1024, device=dev)
    for _ in range(4):
        assert ops.gemm_dx(dY, W) is not None
elif sec == 'dw':
    dY = torch.randn(64 * 1024, 1024, device=dev)
    X = torch.randn(64 * 1024, 2784, device=dev)
    for _ in range(4):
        assert ops.gemm_dw(dY, X) is not None
elif sec == 'attn_core':
    B, R, nh, HP = 64, 1000, 6, 176
    Rp = 1024
    qkv = torch.zeros(B, Rp, 3, nh, HP, device=dev)
    qkv[..., :171] = torch.randn(B, Rp, 3, nh, 171, device=dev) * 0.5
    qkv = qkv.reshape(B, Rp, 3 * nh * HP).requires_grad_(True)
    for _ in range(2):
        o = ops.enc_attn_core(qkv, R, nh, 1.0 / 32, 0.2)
        o.backward(torch.randn_like(o))
        qkv.grad = None
elif sec == 'logit':
    gemm(256, 5000, 1024, 8)
elif sec == 'attn_hid':
    gemm(256, 1024, 1024, 8)
elif sec == 'lstm':
    B, H = 256, 1024
    x, h, c = torch.randn(B, 1536, device=dev), torch.randn(B, H, device=dev), torch.randn(B, H, device=dev)
    w_ih, w_hh = torch.randn(4 * H, 1536, device=dev) * 0.02, torch.randn(4 * H, H, device=dev) * 0.02
    b = torch.zeros(4 * H, device=dev)
    for _ in range(8):
        ops.lstm_cell([x], [w_ih], h, w_hh, b, b, c)
elif sec == 'flash':
    qkv = torch.randn(256, 1000, 18 * ops.HEAD_PAD, device=dev) * 0.5
    qkv.view(256, 1000, 18, ops.HEAD_PAD)[..., 171:] = 0
    for _ in range(3):
        ops.flash_attn_padded(qkv, 6, 1.0 / 32.0)
else:
    raise SystemExit('unknown section ' + sec)
torch.cuda.synchronize()
