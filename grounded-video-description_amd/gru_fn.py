"""Training path of the frame-wise context encoder nn.GRU(1024, 512, 2, bidirectional, batch_first, dropout=0.2)
(model.py:150-154,399) on the HIP kernels, forward AND backward.

Forward of a layer = the inference path: ONE MFMA GEMM for the input projections of both directions + ONE persistent
cooperative kernel for the recurrence (csrc/gru.hip) instead of ~6 library launches per (step, direction).  Backward is
a hand-scheduled BPTT that keeps nothing but the layer's input projections `gi` and its output (= the hidden sequence):
  * gh = h_{t-1} W_hh^T + b_hh for EVERY step at once (one GEMM per direction over the shifted output), gates recomputed;
  * per reverse step ONE pointwise kernel for both directions (gvd_gru_bwd_step) + one small GEMM per direction
    (d_gh W_hh, the recurrent gradient);
  * dX, dW_ih, dW_hh and the bias gradients as GEMMs / reductions over all steps after the loop.
Reference semantics: autograd through torch.nn.GRU (gate order r,z,n; n = tanh(gi_n + r * gh_n)).
"""
import torch
import torch.nn.functional as F

from . import ops

K = ops   # kernel backend; tests substitute the torch stand-ins of tests/torch_backend.py to check the algebra on CPU


def _mm_nt(a, bt, out=None):
    """a [M,K] @ bt[N,K]^T on the MFMA GEMM."""
    return K.gemm_nt(a, bt, out=out)


def _mm_tn(a, b):
    """a[M,N]^T @ b[M,K] -> [N,K] (a weight gradient: contraction over the B*T rows).  The K-strided MFMA kernel takes the
    operands in place when the shape is one it is built for; otherwise both are transposed into zero-padded [., Mp]
    copies (Mp a multiple of the GEMM's 32-deep k tile) for the plain kernel — these products are small."""
    f = getattr(K, 'gemm_dw', None)
    if f is not None and a.is_contiguous() and b.is_contiguous():
        r = f(a, b)
        if r is not None:
            return r
    M = a.shape[0]
    Mp = -(-M // 32) * 32
    at = a.new_zeros(a.shape[1], Mp)
    bt = b.new_zeros(b.shape[1], Mp)
    at[:, :M] = a.t()
    bt[:, :M] = b.t()
    return K.gemm_nt(at, bt)


class GruLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_ih, b_ih, w_hh_f, b_hh_f, w_hh_b, b_hh_b, flags):
        B, T, In = x.shape
        Hh = w_hh_f.shape[1]
        x = x.contiguous()
        gi = K.gemm_nt(x.view(B * T, In), w_ih, b_ih)                                  # [B*T, 2*3*Hh]
        out = K.gru_layer(gi, w_hh_f.contiguous(), b_hh_f.contiguous(), w_hh_b.contiguous(), b_hh_b.contiguous(),
                          B, T, Hh, flags)                                             # [B,T,2*Hh]
        ctx.save_for_backward(x, w_ih, w_hh_f, b_hh_f, w_hh_b, b_hh_b, gi, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w_ih, w_hh_f, b_hh_f, w_hh_b, b_hh_b, gi, out = ctx.saved_tensors
        B, T, In = x.shape
        Hh = w_hh_f.shape[1]
        dev, dt = x.device, x.dtype
        dout = dout.contiguous()
        w_hh, b_hh = (w_hh_f, w_hh_b), (b_hh_f, b_hh_b)
        # h_{t-1} of every step and direction (forward direction: the output one step earlier; backward: one later)
        hprev = torch.zeros(B, T, 2, Hh, device=dev, dtype=dt)
        if T > 1:
            hprev[:, 1:, 0] = out[:, :-1, :Hh]
            hprev[:, :-1, 1] = out[:, 1:, Hh:]
        gh = torch.empty(B, T, 2, 3 * Hh, device=dev, dtype=dt)
        for d in range(2):
            K.gemm_nt(hprev[:, :, d].reshape(B * T, Hh), w_hh[d], b_hh[d], out=gh.view(B * T, 2, 3 * Hh)[:, d])
        d_gi = torch.empty(B, T, 2, 3 * Hh, device=dev, dtype=dt)
        d_gh = torch.empty(B, T, 2, 3 * Hh, device=dev, dtype=dt)
        carry_z = torch.empty(2, B, Hh, device=dev, dtype=dt)
        carry_mm = torch.empty(2, B, Hh, device=dev, dtype=dt)
        w_hh_t = [w.t().contiguous() for w in w_hh]                                    # [Hh, 3*Hh]
        for i in range(T):
            tf, tb = T - 1 - i, i
            K.gru_bwd_step(dout, gi, gh, out, carry_mm, carry_z, d_gi, d_gh, B, T, Hh, tf, tb, i == 0)
            if i + 1 < T:
                _mm_nt(d_gh[:, tf, 0], w_hh_t[0], out=carry_mm[0])
                _mm_nt(d_gh[:, tb, 1], w_hh_t[1], out=carry_mm[1])
        g = [None] * 8
        dgi2 = d_gi.view(B * T, 6 * Hh)
        x2 = x.view(B * T, In)
        if ctx.needs_input_grad[0]:
            g[0] = _mm_nt(dgi2, w_ih.t().contiguous()).view(B, T, In)
        g[1] = _mm_tn(dgi2, x2)                                                        # dW_ih [6*Hh, In]
        g[2] = dgi2.sum(0)
        for d in range(2):
            dgh_d = d_gh[:, :, d].reshape(B * T, 3 * Hh)
            g[3 + 2 * d] = _mm_tn(dgh_d, hprev[:, :, d].reshape(B * T, Hh))
            g[4 + 2 * d] = dgh_d.sum(0)
        return tuple(g)


def gru_bidir_2layer_train(x, gru, flags=None):
    """Differentiable forward of `gru` (nn.GRU, bidirectional, batch_first) over x [B,T,In] -> [B,T,2*Hh]; inter-layer
    dropout as in nn.GRU when the module is in training mode."""
    inp = x
    for l in range(gru.num_layers):
        g = lambda n: getattr(gru, '%s_l%d' % (n, l))
        gr = lambda n: getattr(gru, '%s_l%d_reverse' % (n, l))
        w_ih = torch.cat([g('weight_ih'), gr('weight_ih')], 0)
        b_ih = torch.cat([g('bias_ih'), gr('bias_ih')], 0)
        out = GruLayerFn.apply(inp, w_ih, b_ih, g('weight_hh'), g('bias_hh'), gr('weight_hh'), gr('bias_hh'), flags)
        if l + 1 < gru.num_layers and gru.training and gru.dropout > 0:
            out = K.dropout(out, gru.dropout, True) if hasattr(K, 'dropout') and out.is_cuda else F.dropout(out, gru.dropout, True)
        inp = out
    return inp
