"""Training path of the frame-wise context encoder under `--t_attn_mode bilstm` (opts.py:60): nn.LSTM(1024, 512, 2,
bidirectional, batch_first, dropout=0.2) (model.py:145-149,399) on the HIP kernels, forward AND backward.

Forward of a layer: ONE MFMA GEMM for the input projections of both directions + ONE persistent kernel for the recurrence
(csrc/lstm_seq.hip), which also keeps the post-activation gates and the cell state of every step.  Backward is a
hand-scheduled BPTT: per reverse step and direction the pointwise LSTM-cell backward kernel the decoder's BPTT uses
(gvd_lstm_cell_bwd: from d h, d c and the saved gates / cell states to the gate gradients and d c_{t-1}) and the recurrent
product d h_{t-1} = d gates W_hh; dX, dW_ih, dW_hh and the bias gradients as GEMMs / reductions over all steps after the
loop.  Reference semantics: autograd through torch.nn.LSTM (gate order i,f,g,o; b_ih and b_hh both present).
"""
import torch
import torch.nn.functional as F

from . import gru_fn, ops
from .decoder_bwd import _dx

K = ops   # kernel backend; tests substitute the torch stand-ins of tests/torch_backend.py to check the algebra on CPU


class LstmLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_ih, b_ih, w_hh_f, b_hh_f, w_hh_b, b_hh_b, flags):
        B, T, In = x.shape
        Hh = w_hh_f.shape[1]
        x = x.contiguous()
        gi = K.gemm_nt(x.view(B * T, In), w_ih, b_ih)                                  # [B*T, 2*4*Hh]
        out, gates, c_seq = K.lstm_seq_layer(gi, w_hh_f.contiguous(), b_hh_f.contiguous(), w_hh_b.contiguous(),
                                             b_hh_b.contiguous(), B, T, Hh, flags, save=True)
        ctx.save_for_backward(x, w_ih, w_hh_f, w_hh_b, out, gates, c_seq)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w_ih, w_hh_f, w_hh_b, out, gates, c_seq = ctx.saved_tensors
        B, T, In = x.shape
        Hh = w_hh_f.shape[1]
        dev, dt = x.device, x.dtype
        dout = dout.contiguous()
        w_hh = (w_hh_f, w_hh_b)
        d_g = torch.empty(B, T, 2, 4 * Hh, device=dev, dtype=dt)          # gate gradients = d gi = d (h_{t-1} W_hh^T + b_hh)
        zero = torch.zeros(B, Hh, device=dev, dtype=dt)
        dh_rec = [None, None]                                              # d h_t from step t -+ 1: d gates W_hh
        dh_buf = [[torch.empty(B, Hh, device=dev, dtype=dt) for _ in range(2)] for _ in range(2)]     # ping-pong per direction
        dc_next = [None, None]
        for i in range(T):
            first = i == T - 1                                             # each direction's own first step: c_{t-1} = 0
            groups = []
            for d, t in ((0, T - 1 - i), (1, i)):                         # forward direction walks T-1..0, backward 0..T-1
                tp = t - 1 if d == 0 else t + 1
                c_prev = zero if first else c_seq[:, tp, d]
                dg, dc_next[d] = K.lstm_cell_bwd(dout[:, t, d * Hh:(d + 1) * Hh], dc_next[d], gates[:, t, d], c_prev,
                                                 c_seq[:, t, d], dg_out=d_g[:, t, d], dh2=dh_rec[d])
                dh_rec[d] = dh_buf[d][i & 1]
                groups.append(dict(A=dg, W=w_hh[d], out=dh_rec[d]))
            if not first:
                _dx(K, groups, B)                                          # both directions' d h_{t-1} = d gates W_hh: one launch
        g = [None] * 8
        dg2 = d_g.view(B * T, 8 * Hh)
        x2 = x.view(B * T, In)
        # h_{t-1} of every step and direction (forward direction: the output one step earlier; backward: one later)
        hprev = torch.zeros(B, T, 2, Hh, device=dev, dtype=dt)
        if T > 1:
            hprev[:, 1:, 0] = out[:, :-1, :Hh]
            hprev[:, :-1, 1] = out[:, 1:, Hh:]
        if ctx.needs_input_grad[0]:
            g[0] = gru_fn._mm_nt(dg2, w_ih.t().contiguous()).view(B, T, In)
        g[1] = gru_fn._mm_tn(dg2, x2)                                      # dW_ih [8*Hh, In]
        g[2] = dg2.sum(0)
        for d in range(2):
            dgd = d_g[:, :, d].reshape(B * T, 4 * Hh)
            g[3 + 2 * d] = gru_fn._mm_tn(dgd, hprev[:, :, d].reshape(B * T, Hh))
            g[4 + 2 * d] = dgd.sum(0)
        return tuple(g)


def lstm_bidir_2layer_train(x, lstm, flags=None):
    """Differentiable forward of `lstm` (nn.LSTM, bidirectional, batch_first) over x [B,T,In] -> [B,T,2*Hh]; inter-layer
    dropout as in nn.LSTM when the module is in training mode."""
    inp = x
    for l in range(lstm.num_layers):
        g = lambda n: getattr(lstm, '%s_l%d' % (n, l))
        gr = lambda n: getattr(lstm, '%s_l%d_reverse' % (n, l))
        w_ih = torch.cat([g('weight_ih'), gr('weight_ih')], 0)
        b_ih = torch.cat([g('bias_ih'), gr('bias_ih')], 0)
        # (b_ih and b_hh enter the gates as a sum: b_hh's gradient equals the gate-gradient column sums, like b_ih's)
        out = LstmLayerFn.apply(inp, w_ih, b_ih, g('weight_hh'), g('bias_hh'), gr('weight_hh'), gr('bias_hh'), flags)
        if l + 1 < lstm.num_layers and lstm.training and lstm.dropout > 0:
            out = K.dropout(out, lstm.dropout, True) if hasattr(K, 'dropout') and out.is_cuda else F.dropout(out, lstm.dropout, True)
        inp = out
    return inp
