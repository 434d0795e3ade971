"""Masked-proposal compaction of the TRAINING step (opt-in: Trainer(compact_rows=True) or GVD_TRAIN_COMPACT=1; see the status note below).

The loader zeroes every proposal whose detection score is at or below `prop_thresh` (features AND box; dataloader_anet.py:
343-344) and marks it in `pnt_mask`.  The reference still pushes all R = T x P rows of a segment through fc7, the class
similarity, pool_embed, the two encoder layers and ctx2pool (model.py:311-391), although
  * every masked row of a segment is the SAME row at every stage (same zero input, row-wise ops; in the encoder a query's
    output depends on the query row and the key SET only), and
  * no loss ever sees a masked row: the region attention of the token loop, `att2_weights` and the grounding logits are
    filled with -1e8 under `pnt_mask` (AttModel.py:98-101, model.py:262-278), the IoU targets of a zero box are zero
    (utils.py:299-328), so the class loss never selects it (model.py:345-350).
The only way a masked row reaches a loss is as a KEY of the encoder's self-attention - and n identical keys are one key whose
score carries + log n (the weighted representative key the inference preamble already uses, csrc/compact.hip).

So the training step can run on a COMPACTED DENSE layout: per segment its valid rows (original order), then ONE
representative masked row with key weight log(n_masked), then pad slots (copies of the same zero row, key weight -inf), the
batch padded to a common Rc = a multiple of 32.  Everything downstream is the ordinary dense training path on [B, Rc, .]
tensors with `pnt_mask` marking representative + pads: same losses (a -1e8 entry contributes exp(.) = 0 to every softmax it
sits in), same parameter gradients (the representative key receives the sum of the n rows' key gradients; its own row as a
query feeds only masked places).  At the synthetic 20 % masking rate Rc = 832 of R = 1000: 17 % fewer rows in every row-wise
GEMM of the step and 31 % smaller attention maps.

TRAIN-MODE CAVEAT (why this stays an opt-in): with dropout live, the reference draws an independent mask for each of the n
masked rows (ctx2pool_grd / loc_fc / pool_embed dropout, the encoder's branch dropouts), so they are no longer identical rows;
here ONE draw stands for all n (weighted n-fold as a key).  Measured on the device (tools/compact_dropout_study.py, 300
train-mode draws per layout, profiles/r04/compact_dropout_study_e.json): the EXPECTED gradient of every parameter agrees
within the Monte-Carlo error (bias / error ratio: median 0.99, max 1.91 over the 81 parameters), the mean losses within
0.4 sigma - no bias is detectable; the stochastic regulariser is still a different one in second order, and a training run that
must reproduce the reference's statistics to the letter keeps the full row set.

STATUS (round 4): the index construction and the loss / gradient equivalence are pinned on the CPU against the oracle
(tests/test_train_compact_cpu.py); on the device every training test file passes with the layout on (all mle_* reference
goldens incl. BN train mode, the configuration without the encoder, B = 32 / 64, the 8 x 32 data-parallel case, the optimisation
step, the 2-rank tests: profiles/r04/train_tests_compact_a.txt; a knob-matrix entry keeps it that way).  +21 % segments/s at
batch_size = 64 (866 vs 721).  Switch: train.Trainer(model, opt, compact_rows=True) or GVD_TRAIN_COMPACT=1.
"""
import math

import torch


def compact_regions(ppls, ppls_feat, pnt_mask, frm_mask, min_gain=32, bias_dtype=torch.float32):
    """-> None when compaction does not pay (Rc + min_gain > R), else a dict with the compacted inputs:
    ppls [B,Rc,7], ppls_feat [B,Rc,F], pnt_mask u8 [B,Rc+1], frm_mask [B,Rc,K], key_bias f32 [B,Rc] (0 valid, log n_masked
    on the representative, -inf on the pads), src i64 [B,Rc] (source row of every slot), n_valid i64 [B].
    One device->host read (the batch's largest valid-row count fixes the shapes).  bias_dtype: float64 for the fp64 oracle
    test (log n in fp32 is a 6e-8 relative perturbation of the key weight - rounding level for the fp32 product)."""
    B, R = ppls.shape[0], ppls.shape[1]
    masked = pnt_mask[:, 1:] != 0                                    # [B,R]
    n_valid = (~masked).sum(1)                                       # [B]
    Rc = -(-(int(n_valid.max()) + 1) // 32) * 32                     # the step's one extra host read
    if Rc + min_gain > R:
        return None
    # stable partition: valid rows first in their original order, masked rows after them
    order = torch.sort(masked.to(torch.int64), dim=1, stable=True).indices           # [B,R]
    j = torch.arange(Rc, device=ppls.device).unsqueeze(0)                              # [1,Rc]
    nv = n_valid.unsqueeze(1)
    # (Rc < R and n_valid <= Rc - 1: every segment has at least one masked row, and at least one slot for it)
    first_masked = order.gather(1, nv)                                                 # [B,1]
    src = torch.where(j < nv, order[:, :Rc], first_masked.expand(B, Rc))
    flat = (src + torch.arange(B, device=ppls.device).unsqueeze(1) * R).reshape(-1)

    def rows(x):
        return x.reshape(B * R, *x.shape[2:]).index_select(0, flat).reshape(B, Rc, *x.shape[2:])
    pm = torch.cat([torch.zeros(B, 1, dtype=torch.uint8, device=ppls.device), (j >= nv).to(torch.uint8)], 1)
    n_masked = (R - n_valid).to(bias_dtype).unsqueeze(1)
    neg_inf = torch.full((), -math.inf, dtype=bias_dtype, device=ppls.device)
    zero = torch.zeros((), dtype=bias_dtype, device=ppls.device)
    key_bias = torch.where(j < nv, zero, torch.where(j == nv, torch.log(n_masked), neg_inf)).contiguous()
    return dict(ppls=rows(ppls), ppls_feat=rows(ppls_feat), pnt_mask=pm.contiguous(), frm_mask=rows(frm_mask),
                key_bias=key_bias, src=src, n_valid=n_valid, Rc=Rc)
