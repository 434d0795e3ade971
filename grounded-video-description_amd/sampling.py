"""Multinomial sampling decode (`sample_max=0`, model.py:595-604) on the HIP path.

Per step the reference draws the next word from exp(logprobs / temperature) with torch.multinomial, records the
log-probability of the drawn word (un-tempered, model.py:603) and — unlike the greedy branch — applies no UNK rule.
The TopDownCore step runs on the same HIP kernels as greedy / beam decode (fused LSTM cells, one GEMM for both
attention queries, one streaming pass for both attentions, MFMA logit GEMM + fused log-softmax); only the draw itself
is torch's device-side multinomial, so the token stream depends on the GPU RNG state (torch.manual_seed) and parity
with the CPU reference is distributional, not bitwise (tests/test_gpu_e2e.py::test_multinomial_sampling_statistics).
"""
import torch

from . import ops
from .beam import _core_rows, _state


def multinomial_decode(model, pre, P, temperature=1.0):
    """-> seq i64 [B,L], seqLogprobs f32 [B,L], att2_weights f32 [B,L,R] (masked attention logits, as in greedy)."""
    fc = pre['fc']
    B, H = fc.shape
    R = pre['pool'].shape[1]
    L = model.seq_length
    dev = fc.device
    P = dict(P)
    P['w_stack'] = torch.cat([P['att1_h2att_w'], P['att2_h2att_w']], 0)
    P['b_stack'] = torch.cat([P['att1_h2att_b'], P['att2_h2att_b']], 0)
    fc_gates = ops.gemm_nt(fc, P['att_w_ih'][:, :H], P['att_b_ih']) + P['att_b_hh']
    pm = pre['pnt_mask']
    st = _state(torch.zeros(4, B, H, device=dev))
    seq = torch.empty(B, L, dtype=torch.int64, device=dev)
    lps = torch.empty(B, L, device=dev)
    att2 = torch.empty(B, L, R, device=dev)
    it = torch.zeros(B, dtype=torch.int64, device=dev)                       # <bos> (model.py:588)
    for t in range(L):
        st = _core_rows(P, st, ops.embed_relu(it, P['embed']), fc_gates, pre, pm, 0, att2[:, t])
        logits = ops.gemm_nt(st['h_lang'], P['logit_w'], P['logit_b'])
        lse = ops.logsoftmax_rows(logits)[0]
        logprobs = logits - lse.unsqueeze(1)
        prob = torch.exp(logprobs if temperature == 1.0 else logprobs / temperature)   # model.py:596-600
        it = torch.multinomial(prob, 1).view(-1)
        seq[:, t] = it
        lps[:, t] = logprobs.gather(1, it.view(-1, 1)).view(-1)
    return seq, lps, att2
