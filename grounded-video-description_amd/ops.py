"""Python-side operators of the hot path: thin wrappers that marshal torch tensors into the C-ABI of
libgvd_hip.so (include/gvd_hip.h).  All compute happens in the HIP library; torch only provides
device memory and the current stream.  No CPU/eager fallback exists: non-GPU tensors raise.
"""
import ctypes as C
import os

import torch

from .hip import (AttnSide, DxGroup, GemmArgs, GemmSeg, GreedyArgs, GvdHipError, LstmArgs, check, lib, ptr, require_cuda_f32,
                  stream_ptr)

# ---------------------------------------------------------------------------------------------------------------------
# "No library GEMM on the hot path" as an invariant.  Every place where a product or softmax of the path would leave
# libgvd_hip.so for an ATen / rocBLAS op - a shape one of the kernels does not take - goes through library_fallback()
# first: it is COUNTED always (`library_calls`; bench.py prints the total as "library_gemms") and RAISES under GVD_STRICT=1
# (tests/conftest.py and bench.py switch that on), so a silent fallback cannot survive a test run or a benchmark.
# Without GVD_STRICT the fallback computes the reference's arithmetic through torch, as before.
# ---------------------------------------------------------------------------------------------------------------------
STRICT = os.environ.get('GVD_STRICT', '0') == '1'
library_calls = {}


def set_strict(on):
    global STRICT
    STRICT = bool(on)


def library_fallback(site, detail=''):
    library_calls[site] = library_calls.get(site, 0) + 1
    if STRICT:
        raise GvdHipError('GVD_STRICT: %s would run on a torch library op instead of a kernel of libgvd_hip.so%s'
                          % (site, (' (%s)' % detail) if detail else ''))


def library_call_count():
    return sum(library_calls.values())


def _seg(A, W, K=None, a_bs=0, w_bs=0):
    """A: [..., M, K] view with unit inner stride; W: [..., N, K] likewise."""
    assert A.stride(-1) == 1 and W.stride(-1) == 1
    K = A.shape[-1] if K is None else K
    assert W.shape[-1] == K, (A.shape, W.shape)
    return GemmSeg(ptr(A), A.stride(-2), a_bs, ptr(W), W.stride(-2), w_bs, K)


def gemm_nt(A, W, bias=None, act=0, out=None, m_dev=None, a_row_map=None):
    """out[M,N] = act(A[M,K] @ W[N,K]^T + bias).  nn.Linear forward on the fp32 matrix cores.
    m_dev: optional device int32 tensor (1 element) with the live row count (<= M): rows past it are not computed.
    a_row_map: optional device int32 [M]: output row m reads row a_row_map[m] of A (row gather fused into the operand
    loads); the result then has a_row_map.numel() rows."""
    require_cuda_f32(A, W, bias)
    lead = A.shape[:-1]
    A2 = A.reshape(-1, A.shape[-1])
    if A2.stride(-1) != 1:
        A2 = A2.contiguous()
    M, N = A2.shape[0], W.shape[0]
    if a_row_map is not None:
        assert a_row_map.dtype == torch.int32 and a_row_map.is_contiguous()
        M, lead = a_row_map.numel(), (a_row_map.numel(),)
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    g = GemmArgs()
    g.nseg = 1
    g.seg[0] = _seg(A2, W)
    g.nbias = ptr(bias)
    g.C = ptr(out); g.ldc = out.stride(0)
    g.M, g.N, g.batch, g.act = M, N, 1, act
    g.m_dev = ptr(m_dev)
    if a_row_map is not None:
        g.a_row_map, g.a_src_rows = ptr(a_row_map), A2.shape[0]
    check(lib().gvd_gemm_nt_f32(C.byref(g), stream_ptr()), 'gvd_gemm_nt_f32')
    return out.view(*lead, N)


def gemm_dx(dY, W, addend=None, out=None):
    """dX[M,K] = dY[M,N] @ W[N,K] (+ addend[M,K]) (backward of y = x W^T w.r.t. x) on the pipelined MFMA kernel: W is
    consumed in place as a K-strided operand; `addend` (contiguous: the gradient the same tensor receives through its other
    consumer - a residual connection) enters the product's epilogue as 16-byte reads instead of a separate elementwise
    pass.  Falls back (returns None) when the shape is outside what the kernel takes."""
    M, N = dY.shape
    K = W.shape[1]
    tiles = ((M + 127) // 128) * ((K + 127) // 128)
    if N % 32 or K % 4 or tiles < 256 or not (dY.is_contiguous() and W.is_contiguous()):
        return None
    if addend is not None and not (addend.is_contiguous() and addend.shape == (M, K) and addend.data_ptr() % 16 == 0):
        return None
    out = torch.empty(M, K, device=dY.device, dtype=torch.float32) if out is None else out
    assert out.shape == (M, K) and out.is_contiguous()
    g = GemmArgs()
    g.nseg = 1
    g.seg[0] = GemmSeg(ptr(dY), N, 0, ptr(W), K, 0, N)
    g.C = ptr(out); g.ldc = K
    g.M, g.N, g.batch, g.act = M, K, 1, 0
    g.w_kstrided = 1
    if addend is not None:
        g.rowbias = ptr(addend); g.rowbias_ld = K
    check(lib().gvd_gemm_nt_f32(C.byref(g), stream_ptr()), 'gvd_gemm_nt_f32(dX)')
    return out


def gemm_dx_small(dY, W, addend=None, out=None):
    """dX[M,K] = dY[M,N] @ W[N,K] for the shapes gemm_dx leaves (fewer than 256 output tiles: the token loop's [Lc B, .]
    products, e.g. the vocabulary head's [1280, 5000] x [5000, 1024]) on the grouped small-M kernel (csrc/gemm_dxs.hip).  Its
    contraction runs in 128-deep slices: N is cut into the leading multiple of 128, consumed in place, and a tail (V = 5000:
    8 columns) copied into zero-padded [M, 128] / [128, K] operands whose product enters the main launch as its addend.
    Operands the kernel cannot read in place (rows off 16 bytes, an output width off its 128-column tile) are copied into
    zero-padded ones first; None only for an addend of the wrong shape."""
    M, N = dY.shape
    K = W.shape[1]
    if M < 1:
        return None
    if addend is not None and not (addend.shape == (M, K) and addend.stride(1) == 1):
        return None
    if (N < 128 or dY.stride(1) != 1 or W.stride(1) != 1 or dY.stride(0) % 4 or W.stride(0) % 4 or dY.data_ptr() % 16
            or W.data_ptr() % 16):
        # rows that do not start on 16 bytes (an odd vocabulary size: dY [B Lc, V]) or a contraction shorter than one 128-deep
        # slice: zero-padded copies of both operands (the weight copy is V x 1024 floats: ~10 us) - never the library
        Np = max(128, -(-N // 128) * 128)
        d2 = dY.new_zeros(M, Np)
        d2[:, :N] = dY
        w2 = W.new_zeros(Np, K)
        w2[:N] = W
        return gemm_dx_small(d2, w2, addend, out)
    if K % 128:
        # an output width that is not a multiple of the kernel's 128-column wave tile (the packed wo of the encoder: 1056; the
        # zero-padded fc_embed: 3136 - both only at batch sizes too small for the pipelined kernel): product against the weight
        # zero-padded to the next multiple, the live columns copied out
        Kp = -(-K // 128) * 128
        Wp = W.new_zeros(N, Kp)
        Wp[:, :K] = W
        full = gemm_dx_small(dY, Wp)
        if full is None:
            return None
        r = full[:, :K] if addend is None else full[:, :K] + addend
        return r.contiguous() if out is None else out.copy_(r)
    N0 = N - N % 128
    out = torch.empty(M, K, device=dY.device, dtype=torch.float32) if out is None else out
    tail = addend                          # (`addend` [M,K]: a further term of the result, e.g. a residual path's gradient)
    if N0 < N:
        a = dY.new_zeros(M, 128)
        a[:, :N - N0] = dY[:, N0:]
        w = W.new_zeros(128, K)
        w[:N - N0] = W[N0:]
        tail = torch.empty(M, K, device=dY.device, dtype=torch.float32)
        dx_products([dict(A=a, W=w, out=tail, addend=addend)], M)
    dx_products([dict(A=dY[:, :N0], W=W[:N0], out=out, addend=tail)], M)
    return out


def dx_any(dY, W, addend=None, out=None, what='dX'):
    """dY @ W (+ addend) on whichever kernel takes the shape: the pipelined K-strided product (>= 256 output tiles), the
    grouped small-M kernel, else - counted, and an error under GVD_STRICT - the library."""
    r = gemm_dx(dY, W, addend, out)
    if r is None:
        r = gemm_dx_small(dY, W, addend, out)
    if r is None:
        library_fallback(what, '%s x %s' % (tuple(dY.shape), tuple(W.shape)))
        r = dY @ W if addend is None else torch.addmm(addend, dY, W)
        if out is not None:
            r = out.copy_(r)
    return r


def dw_any(dY, X, what='dW'):
    """dY^T @ X on the K-strided MFMA kernel, else - counted, an error under GVD_STRICT - the library."""
    r = gemm_dw(dY, X)
    if r is None:
        library_fallback(what, '%s^T x %s, strides %s' % (tuple(dY.shape), tuple(X.shape), X.stride()))
        r = dY.t() @ X
    return r


def gemm_dw(dY, X, split=None):
    """dW[N,K] = dY[M,N]^T @ X[M,K] (backward of y = x W^T w.r.t. W): both operands K-strided; the long contraction over
    the M = B*R rows is cut into S chunks run as a batch (deterministic split-K), partial slabs summed.  None = shape not
    taken.  split: force S (tools/dw_split_sweep.py)."""
    M, N = dY.shape
    K = X.shape[1]
    if K % 4 or X.shape[0] != M or M < 1:
        return None
    # What the kernel consumes in place: dY contiguous with N % 4 == 0, X with 16-byte rows (it may be a column block of a
    # wider tensor: att_embed reads segs_feat[:, :, :2048] in place), a contraction length M that is a multiple of its 32-deep
    # k tile.  Anything else (the token loop's B Lc rows at odd batch sizes, an odd vocabulary size, B R % 32 != 0) is copied
    # into zero-padded operands first - zeros add nothing to the sums - instead of leaving the library of kernels.
    N_real = N
    pad_m, pad_n = (-M) % 32, (-N) % 4
    if pad_m or pad_n or not dY.is_contiguous() or dY.data_ptr() % 16:
        d2 = dY.new_zeros(M + pad_m, N + pad_n)
        d2[:M, :N] = dY
        dY, N = d2, N + pad_n
    if pad_m or X.stride(1) != 1 or X.stride(0) < K or X.stride(0) % 4 or X.data_ptr() % 16:
        x2 = X.new_zeros(M + pad_m, K)
        x2[:M] = X
        X = x2
    M += pad_m
    ldx = X.stride(0)
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    S = 1
    while tiles * S < 1024 and M % (64 * S) == 0 and M // (2 * S) >= 2048:
        S *= 2
    if (tiles * S) % 512:
        # tile counts that do not fill the 512 resident workgroup slots evenly (pool_embed 176, q|k|v 200, wo 72 tiles):
        # more, shorter workgroups even the occupancy out - measured (profiles/r03/dw_split_sweep_z3.log) q|k|v 3.54 ->
        # 3.26 ms at S = 16, pool_embed 2.94 -> 2.70 ms at S = 16, wo 1.26 -> 1.16 ms at S = 32; shapes whose tiles x S is a
        # multiple of 512 (fc7, ff1, ff2, ctx2pool) are flat in S and keep the smaller slab count
        while tiles * S < 2304 and S < 32 and M % (64 * S) == 0 and M // (2 * S) >= 1024:
            S *= 2
    if split is not None:
        S = split
    if tiles * S < 256:
        # few output tiles and a short contraction (the [Lc B, .] weight gradients of the token loop: M = 640 .. 1280): cut the
        # contraction down to 64-row chunks if that is what it takes to give every CU a workgroup
        while tiles * S < 256 and M % (64 * S) == 0 and M // (2 * S) >= 32:
            S *= 2
    if M % (32 * S):
        return None
    Mc = M // S
    part = torch.empty(S, N, K, device=dY.device, dtype=torch.float32)
    g = GemmArgs()
    g.nseg = 1
    g.seg[0] = GemmSeg(ptr(dY), N, Mc * N, ptr(X), ldx, Mc * ldx, Mc)
    g.C = ptr(part); g.ldc = K; g.c_batch_stride = N * K
    g.M, g.N, g.batch, g.act = N, K, S, 0
    g.a_kstrided = g.w_kstrided = 1
    check(lib().gvd_gemm_nt_f32(C.byref(g), stream_ptr()), 'gvd_gemm_nt_f32(dW)')
    out = part[0] if S == 1 else part.sum(0)
    return out if N == N_real else out[:N_real].contiguous()


def grounder_dot(xt, feats, mask, mbias=None, rowbias=None, xt_shared=False):
    """`AttModel._grounder` dot-product branch (model.py:243-280), batched over B in one launch:
    out[b,m,r] = xt[b,m,:] . feats[b,r,:] + mbias[(b,)m] + rowbias[b,m,r];  out[mask] = -1e8.
    xt: [B,M,K] (or [M,K] with xt_shared, e.g. the D1 visual words); feats: [B,R,K];
    mask: u8 [B,R] (broadcast over m) or [B,M,R]."""
    require_cuda_f32(xt, feats, mbias, rowbias)
    B, R, K = feats.shape
    M = xt.shape[-2]
    out = torch.empty(B, M, R, device=feats.device, dtype=torch.float32)
    g = GemmArgs()
    g.nseg = 1
    assert xt.is_contiguous() and feats.is_contiguous()
    g.seg[0] = GemmSeg(ptr(xt), K, 0 if xt_shared else M * K, ptr(feats), K, R * K, K)
    if mbias is not None:
        assert mbias.is_contiguous()
        g.mbias = ptr(mbias)
        g.mbias_batch_stride = 0 if mbias.dim() == 1 else M
    if rowbias is not None:
        assert rowbias.is_contiguous() and rowbias.shape == (B, M, R)
        g.rowbias = ptr(rowbias); g.rowbias_ld = R; g.rowbias_batch_stride = M * R
    if mask is not None:
        assert mask.dtype == torch.uint8 and mask.stride(-1) == 1
        g.mask = ptr(mask)
        if mask.dim() == 2:
            g.mask_ldm = 0; g.mask_batch_stride = mask.stride(0)
        else:
            g.mask_ldm = mask.stride(1); g.mask_batch_stride = mask.stride(0)
    g.C = ptr(out); g.ldc = R; g.c_batch_stride = M * R
    g.M, g.N, g.batch, g.act = M, R, B, 0
    check(lib().gvd_gemm_nt_f32(C.byref(g), stream_ptr()), 'gvd_gemm_nt_f32(grounder)')
    return out


def _mask_strides(mask, M):
    """(ptr, ld_mask, batch stride) of a u8 mask [B,R] (one row per sample) or [B,M,R] with unit inner stride."""
    if mask is None:
        return None, 0, 0
    assert mask.dtype == torch.uint8 and mask.stride(-1) == 1
    if mask.dim() == 2:
        return ptr(mask), 0, mask.stride(0)
    assert mask.shape[1] == M
    return ptr(mask), mask.stride(1), mask.stride(0)


def grounder_stream(xt, feats, mask, mbias=None, rowbias=None):
    """`AttModel._grounder` dot branch (model.py:262-278) for the few words of a caption against the segment's region
    features - the HBM-streaming kernel (gvd_grounder_fwd_f32): out[b,m,r] = xt[b,m,:] . feats[b,r,:] + mbias[b,m] +
    rowbias[b,m,r]; out[mask] = -1e8.  xt [B,M,K] (M <= 32), feats [B,R,K] (K % 32 == 0), mask u8 [B,R] | [B,M,R]."""
    require_cuda_f32(xt, feats, mbias, rowbias)
    B, R, K = feats.shape
    M = xt.shape[1]
    assert xt.is_contiguous() and feats.is_contiguous() and xt.shape == (B, M, K)
    out = torch.empty(B, M, R, device=feats.device, dtype=torch.float32)
    mp, mld, mbs = _mask_strides(mask, M)
    if mbias is not None:
        assert mbias.is_contiguous() and mbias.shape == (B, M)
    if rowbias is not None:
        assert rowbias.is_contiguous() and rowbias.shape == (B, M, R)
    check(lib().gvd_grounder_fwd_f32(ptr(feats), K, R * K, ptr(xt), K, M * K, ptr(mbias), M, ptr(rowbias), R, M * R,
                                     mp, mld, mbs, ptr(out), R, M * R, B, M, R, K, stream_ptr()), 'gvd_grounder_fwd_f32')
    return out


def grounder_stream_any(xt, feats, mask, mbias=None, rowbias=None):
    """grounder_stream for any number of words M (32-word chunks above 32: `--seq_length 40`, README.md:115)."""
    M = xt.shape[1]
    if M <= 32:
        return grounder_stream(xt, feats, mask, mbias, rowbias)
    c = lambda t: None if t is None else t.contiguous()
    return torch.cat([grounder_stream(c(xt[:, m0:m1]), feats, _mchunk(mask, m0, m1, True), c(_mchunk(mbias, m0, m1)),
                                      c(_mchunk(rowbias, m0, m1))) for m0, m1 in _m_chunks(M)], 1)


def rows_contract(S, F, mask=None, S_t=None):
    """out[b,m,:] = sum_r S[b,m,r] F[b,r,:] (entries of S under `mask` count as 0).  S [B,M,R] (M <= 32, unit inner stride),
    F [B,R,N] contiguous (N % 128 == 0) -> [B,M,N]: one streaming pass over F (gvd_rows_contract_f32).  S_t: the transposed,
    already masked copy [B,R,32] masked_copy_rowsum(..., want_t=True) writes - read instead of S when given."""
    require_cuda_f32(S, F, S_t)
    B, M, R = S.shape
    N = F.shape[2]
    assert F.is_contiguous() and F.shape[:2] == (B, R) and S.stride(2) == 1
    assert S_t is None or (S_t.is_contiguous() and S_t.shape == (B, R, 32))
    out = torch.empty(B, M, N, device=F.device, dtype=torch.float32)
    mp, mld, mbs = _mask_strides(mask, M)
    check(lib().gvd_rows_contract_f32(ptr(S), S.stride(1), S.stride(0), mp, mld, mbs, ptr(S_t), ptr(F), N, R * N, ptr(out), N,
                                      M * N, B, M, R, N, stream_ptr()), 'gvd_rows_contract_f32')
    return out


def rank_update(S, X, mask=None):
    """out[b,r,:] = sum_m S[b,m,r] X[b,m,:] (entries of S under `mask` count as 0).  S [B,M,R] (M <= 32, unit inner stride),
    X [B,M,N] (unit inner stride, N % 128 == 0, 16-byte aligned rows) -> [B,R,N], written once (gvd_rank_update_f32)."""
    require_cuda_f32(S, X)
    B, M, R = S.shape
    N = X.shape[2]
    assert X.shape[:2] == (B, M) and S.stride(2) == 1 and X.stride(2) == 1
    out = torch.empty(B, R, N, device=X.device, dtype=torch.float32)
    mp, mld, mbs = _mask_strides(mask, M)
    check(lib().gvd_rank_update_f32(ptr(S), S.stride(1), S.stride(0), mp, mld, mbs, ptr(X), X.stride(1), X.stride(0),
                                    ptr(out), N, R * N, B, M, R, N, stream_ptr()), 'gvd_rank_update_f32')
    return out


def rank_update_any(S, X, mask=None):
    """rank_update for any number of rows M: the kernel holds <= 32 rows of X in registers, longer operands (teacher-forced
    captions above 32 tokens, `--seq_length 40`) are cut into 32-row chunks - one more stream of the output per chunk, still
    no library GEMM."""
    M = S.shape[1]
    if M <= 32:
        return rank_update(S, X, mask)
    out = None
    for m0 in range(0, M, 32):
        mk = mask if (mask is None or mask.dim() == 2) else mask[:, m0:m0 + 32]
        part = rank_update(S[:, m0:m0 + 32], X[:, m0:m0 + 32], mk)
        out = part if out is None else out.add_(part)
    return out


_dx_ws = {}


def dx_products(groups, M):
    """Up to 4 small-M products out = A @ W (+ addend) in ONE launch (gvd_gemm_dx_small_f32; csrc/gemm_dxs.hip): the gradients
    w.r.t. the inputs of the LSTM cells / h2att of one BPTT step.  groups: list of dict(A [M,Kred], W [Kred,ncols] (a column
    block view of a weight: unit inner stride), out [M,ncols] (view, unit inner stride)[, addend [M,ncols]])."""
    assert 1 <= len(groups) <= 4
    arr = (DxGroup * len(groups))()
    total = 0
    for i, g in enumerate(groups):
        A, W, out, add = g['A'], g['W'], g['out'], g.get('addend')
        require_cuda_f32(A, W, out, add)
        Kred, ncols = W.shape
        assert A.shape == (M, Kred) and out.shape == (M, ncols) and A.stride(1) == 1 and W.stride(1) == 1 and out.stride(1) == 1
        arr[i].A, arr[i].lda = ptr(A), A.stride(0)
        arr[i].W, arr[i].ldw = ptr(W), W.stride(0)
        arr[i].Kred, arr[i].ncols = Kred, ncols
        arr[i].out, arr[i].ldo = ptr(out), out.stride(0)
        if add is not None:
            assert add.shape == (M, ncols) and add.stride(1) == 1
            arr[i].addend, arr[i].ld_add = ptr(add), add.stride(0)
        total += ncols
    dev = groups[0]['A'].device
    need = lib().gvd_gemm_dx_small_workspace_bytes(M, total)
    key = (dev.index, torch.cuda.current_stream().cuda_stream)
    ws = _dx_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.zeros(need, dtype=torch.uint8, device=dev)          # tile counters start at zero; every launch re-zeroes them
        _dx_ws[key] = ws
    check(lib().gvd_gemm_dx_small_f32(arr, len(groups), M, ptr(ws), ws.numel(), stream_ptr()), 'gvd_gemm_dx_small_f32')


def dx_ok(M, Kred, ncols):
    return Kred % 128 == 0 and ncols % 128 == 0 and M >= 1


def softmax_rows(x, out=None):
    """softmax over the last axis of x [..., N] (unit inner stride; leading axes collapse to rows of one stride)."""
    require_cuda_f32(x)
    N = x.shape[-1]
    x2 = x.reshape(-1, N)
    assert x2.stride(1) == 1
    out = torch.empty(x2.shape, device=x.device, dtype=torch.float32) if out is None else out.view(-1, N)
    check(lib().gvd_softmax_rows(ptr(x2), x2.stride(0), x2.shape[0], N, ptr(out), out.stride(0), stream_ptr()), 'gvd_softmax_rows')
    return out.view(x.shape)


def lstm_cell(xs, ws, h_prev, w_hh, b_ih, b_hh, c_prev, rowbias=None, gates_out=None, h_out=None, c_out=None):
    """Fused nn.LSTMCell forward.  xs/ws: lists of input blocks [B,K_s] and the matching column blocks
    of weight_ih ([4H,K_s] views, row stride = full weight_ih width).  Returns (h, c); `h_out` / `c_out` ([B,H] views
    with unit inner stride, e.g. one step of the BPTT's saved-state arrays) receive them in place when given."""
    B, H = c_prev.shape
    require_cuda_f32(h_prev, c_prev, w_hh, *xs, *ws)
    h = torch.empty(B, H, device=c_prev.device, dtype=torch.float32) if h_out is None else h_out
    c = torch.empty(B, H, device=c_prev.device, dtype=torch.float32) if c_out is None else c_out
    assert h.stride(-1) == 1 and c.stride(-1) == 1 and h.shape == (B, H) and c.shape == (B, H)
    a = LstmArgs()
    segs = list(zip(xs, ws)) + [(h_prev, w_hh)]
    a.nseg = len(segs)
    for i, (x, w) in enumerate(segs):
        a.seg[i] = _seg(x, w)
    a.b_ih, a.b_hh = ptr(b_ih), ptr(b_hh)
    if rowbias is not None:
        a.rowbias = ptr(rowbias); a.rowbias_ld = rowbias.stride(0)
    a.c_prev = ptr(c_prev); a.ldc_prev = c_prev.stride(0)
    a.h_out = ptr(h); a.ldh = h.stride(0)
    a.c_out = ptr(c); a.ldc_out = c.stride(0)
    if gates_out is not None:
        a.gates_out = ptr(gates_out); a.ldg = gates_out.stride(0)
    a.B, a.H = B, H
    check(lib().gvd_lstm_cell_fwd(C.byref(a), stream_ptr()), 'gvd_lstm_cell_fwd')
    return h, c


SCORE_MODES = {'mix': 0, 'mix_mul': 1, 'dp': 2}          # region_attn_mode -> GVD_SCORE_ADD / MUL / DOT (include/gvd_hip.h)


def _side(feats, p_feats, q, w, alpha_bias, att_mask=None, pnt_mask=None, logits_out=None, scores_out=None,
          group=0, row_map=None, score_mode=0):
    s = AttnSide()
    s.score_mode = score_mode          # 2 ('dp'): no alpha_net - w / alpha_bias are None
    assert feats.is_contiguous() and p_feats.is_contiguous() and q.stride(-1) == 1
    s.feats, s.p_feats = ptr(feats), ptr(p_feats)
    s.q = ptr(q); s.ldq = q.stride(0)
    s.w = ptr(w); s.alpha_bias = ptr(alpha_bias)
    if att_mask is not None:
        assert att_mask.dtype == torch.uint8 and att_mask.stride(-1) == 1
        s.att_mask = ptr(att_mask); s.ld_att_mask = att_mask.stride(0)
    if pnt_mask is not None:
        assert pnt_mask.dtype == torch.uint8 and pnt_mask.stride(-1) == 1
        s.pnt_mask = ptr(pnt_mask); s.ld_pnt_mask = pnt_mask.stride(0)
    if logits_out is not None:
        assert logits_out.stride(-1) == 1
        s.logits_out = ptr(logits_out); s.ld_logits = logits_out.stride(0)
    if scores_out is not None:
        assert scores_out.stride(-1) == 1
        s.scores_out = ptr(scores_out); s.ld_scores = scores_out.stride(0)
    s.N = feats.shape[1]
    if row_map is not None:        # feats / p_feats are flat [rows, .] arrays indexed through row_map [B, N]
        assert row_map.dtype == torch.int32 and row_map.is_contiguous() and feats.dim() == 2
        s.row_map = ptr(row_map)
        s.N = row_map.numel() // q.shape[0]
    s.group = group
    return s


_ws_cache = {}


def _workspace(nbytes, device):
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    t = _ws_cache.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = t
    return t


_kernel_timer = None


def set_kernel_timer(timer):
    """Bracket every `attention_step` launch with the HIP event pairs of `timer` (hip.KernelTimer) — bench.py's live
    roofline measurement for the paths that call the attention kernel from Python (training forward, beam search).
    None switches it off."""
    global _kernel_timer
    _kernel_timer = timer


def attention_step(region, temporal, want_separate=False, out=None, cr_out=None, ct_out=None, sum_region=True):
    """Both additive attentions of one decoder step (AttModel.py:33-53,71-108) in one streaming pass.

    region / temporal: dicts(feats, p_feats, q, w, alpha_bias[, att_mask, pnt_mask, logits_out]).
    Returns att+att2 [B,H] (and the two contexts when want_separate); `out` ([B,H], unit inner stride) and the contiguous
    `cr_out` / `ct_out` receive them in place when given.  att_input_mode (AttModel.py:140-151): temporal = None is
    'region' (out = att2); sum_region = False is 'featmap' (out = att, the frame-wise context alone - the region side still
    runs for its logits / scores and, with want_separate, its context)."""
    f = region['feats']
    B = region['q'].shape[0]                      # rows (= feats.shape[0] * group)
    if region.get('row_map') is not None:
        Nr, H = region['row_map'].numel() // B, f.shape[-1]
    else:
        Nr, H = f.shape[1], f.shape[2]
        assert B == f.shape[0] * max(region.get('group', 0), 1)
    A = region['p_feats'].shape[-1]
    require_cuda_f32(f, region['p_feats'], region['q'])
    sr = _side(**region)
    st = _side(**temporal) if temporal is not None else None
    Nt = temporal['feats'].shape[1] if temporal is not None else 0
    ws = _workspace(lib().gvd_attn_workspace_bytes(B, Nr, Nt, H), f.device)
    if out is None:
        out = torch.empty(B, H, device=f.device, dtype=torch.float32)
    cr = ct = None
    if want_separate:
        cr = torch.empty(B, H, device=f.device, dtype=torch.float32) if cr_out is None else cr_out
        if st is not None:
            # ('featmap': `out` IS the frame-wise context - a separate copy only where the caller asked for one by name)
            ct = ct_out if (ct_out is not None or not sum_region) else torch.empty(B, H, device=f.device, dtype=torch.float32)
    assert out.stride(-1) == 1 and out.shape == (B, H) and all(t is None or t.is_contiguous() for t in (cr, ct))
    prof = _kernel_timer.h if _kernel_timer is not None else None
    if not sum_region:
        # 'featmap': the kernel writes the temporal context where the sum would go (its per-side outputs are contiguous rows)
        assert st is not None and out.is_contiguous()
        check(lib().gvd_attn_fwd_prof(C.byref(sr), C.byref(st), B, A, H, None, 0, ptr(cr), ptr(out), ptr(ws), prof,
                                      stream_ptr()), 'gvd_attn_fwd')
        if ct is not None and ct.data_ptr() != out.data_ptr():
            ct.copy_(out)
        return (out, cr, out if ct is None else ct) if want_separate else out
    check(lib().gvd_attn_fwd_prof(C.byref(sr), C.byref(st) if st is not None else None, B, A, H, ptr(out), out.stride(0),
                                  ptr(cr), ptr(ct), ptr(ws), prof, stream_ptr()), 'gvd_attn_fwd')
    return (out, cr, ct) if want_separate else out


def embed_relu(it, embed):
    B = it.shape[0]
    xt = torch.empty(B, embed.shape[1], device=embed.device, dtype=torch.float32)
    assert it.dtype == torch.int64
    check(lib().gvd_embed_relu(ptr(it), it.stride(0), B, ptr(embed), embed.shape[1], ptr(xt), xt.stride(0),
                               stream_ptr()), 'gvd_embed_relu')
    return xt


def logsoftmax_rows(logits, target=None, topk=0):
    """Per-row logsumexp (+ log-prob of `target`, + top-k log-probs/indices)."""
    require_cuda_f32(logits)
    rows, V = logits.shape
    lse = torch.empty(rows, device=logits.device, dtype=torch.float32)
    picked = torch.empty(rows, device=logits.device, dtype=torch.float32) if target is not None else None
    tv = torch.empty(rows, topk, device=logits.device, dtype=torch.float32) if topk else None
    ti = torch.empty(rows, topk, device=logits.device, dtype=torch.int64) if topk else None
    check(lib().gvd_logsoftmax_rows(ptr(logits), logits.stride(0), rows, V, ptr(lse), ptr(target), ptr(picked),
                                    topk, ptr(tv), ptr(ti), stream_ptr()), 'gvd_logsoftmax_rows')
    return lse, picked, tv, ti


ATT_INPUT_MODES = {'both': 0, 'featmap': 1, 'region': 2}          # GVD_ATT_INPUT_* (include/gvd_hip.h)


def greedy_decode(pre, P, pnt_mask, L, unk_idx, prof=None, flags=None, att_input_mode='both', region_attn_mode='mix'):
    """Whole greedy token loop (AttModel._sample, model.py:580-624) in one C call.
    pre: dict(fc, conv, p_conv, pool, p_pool) from the preamble; P: dict of parameter tensors.
    `flags`: list that receives the launch's device status word (non-zero after a sync = the persistent kernel's grid
    barrier timed out and the ids are poisoned); the caller must check it before trusting the result
    (TopDownModel.check_kernel_status)."""
    fc = pre['fc'].contiguous()
    # att_input_mode='region' (AttModel.py:140-141,151-152): no frame-wise attention, the preamble holds no conv / p_conv
    conv, p_conv = (None, None) if att_input_mode == 'region' else (pre['conv'].contiguous(), pre['p_conv'].contiguous())
    B, H = fc.shape
    row_map = None
    if pre.get('pool') is None:
        # compacted preamble (masked-proposal compaction): the region features are consumed in place through the row map;
        # the decode-batch persistent kernel (B <= 4) reads the dense layout
        if B > 4:
            pool, p_pool, row_map = pre['pool_c'].contiguous(), pre['p_pool_c'].contiguous(), pre['ci'].cidx
        else:
            pool, p_pool = pre['ci'].expand(pre['pool_c']), pre['ci'].expand(pre['p_pool_c'])
    else:
        pool, p_pool = pre['pool'].contiguous(), pre['p_pool'].contiguous()
    require_cuda_f32(fc, pool, p_pool, *([] if conv is None else [conv, p_conv]))
    Ft, R, A = (0 if conv is None else conv.shape[1]), pnt_mask.shape[1] - 1, p_pool.shape[-1]
    V, E = P['embed'].shape
    dev = fc.device
    seq = torch.empty(B, L, dtype=torch.int64, device=dev)
    lps = torch.empty(B, L, dtype=torch.float32, device=dev)
    att2 = torch.empty(B, L, R, dtype=torch.float32, device=dev)
    pm = pnt_mask.contiguous()
    assert pm.dtype == torch.uint8 and pm.shape == (B, R + 1)
    ws = torch.empty(lib().gvd_greedy_workspace_bytes(B, Ft, R, H, A, E, V), dtype=torch.uint8, device=dev)
    a = GreedyArgs()
    a.fc, a.conv, a.p_conv, a.pool, a.p_pool = ptr(fc), ptr(conv), ptr(p_conv), ptr(pool), ptr(p_pool)
    a.pnt_mask = ptr(pm)
    a.pool_row_map = ptr(row_map)
    for k in ('embed', 'att_w_ih', 'att_w_hh', 'att_b_ih', 'att_b_hh', 'lang_w_ih', 'lang_w_hh', 'lang_b_ih',
              'lang_b_hh', 'att1_h2att_w', 'att1_h2att_b', 'att1_alpha_w', 'att1_alpha_b', 'att2_h2att_w',
              'att2_h2att_b', 'att2_alpha_w', 'att2_alpha_b', 'logit_w', 'logit_b'):
        t = P.get(k)
        if t is None and region_attn_mode == 'dp' and k in ('att2_alpha_w', 'att2_alpha_b'):
            continue                  # dot-product region attention: the module has no alpha_net (AttModel.py:63-66,92-95)
        assert t.is_contiguous() and t.is_cuda and t.dtype == torch.float32, k
        setattr(a, k, ptr(t))
    a.B, a.Ft, a.R, a.H, a.A, a.E, a.V, a.L, a.unk_idx = B, Ft, R, H, A, E, V, L, unk_idx
    a.seq, a.seq_logprobs, a.att2_weights, a.workspace = ptr(seq), ptr(lps), ptr(att2), ptr(ws)
    a.prof = prof.h if prof is not None else None
    a.no_persistent = 0 if _persistent['on'] else 1
    a.att_input_mode = ATT_INPUT_MODES[att_input_mode]
    a.region_attn_mode = SCORE_MODES[region_attn_mode]
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    a.status = ptr(status)
    trace = None
    if os.environ.get('GVD_PD_TRACE'):        # profiling aid: phase time stamps of the persistent kernel
        trace = torch.zeros(1 + 7 * L, dtype=torch.int64, device=dev)
        a.trace = ptr(trace)
    greedy_decode.last_trace = trace
    check(lib().gvd_greedy_decode(C.byref(a), stream_ptr()), 'gvd_greedy_decode')
    greedy_decode.last_status = status      # 1 after a device sync = the persistent decode kernel timed out (ids are -1)
    if flags is not None:
        flags.append(status)
    return seq, lps, att2


def iou_targets(ppls, gt_boxes, frm_mask, pnt_mask, want_sim_target=True):
    require_cuda_f32(ppls, gt_boxes)
    B, R, K = frm_mask.shape
    ppls, gt_boxes = ppls.contiguous(), gt_boxes.contiguous()
    frm_mask, pnt_mask = frm_mask.contiguous(), pnt_mask.contiguous()
    ov = torch.empty(B, R, K, device=ppls.device, dtype=torch.float32)
    st = torch.empty(B, K, R, device=ppls.device, dtype=torch.int64) if want_sim_target else None
    check(lib().gvd_iou_targets(ptr(ppls), ppls.shape[2], ptr(gt_boxes), gt_boxes.shape[2], ptr(frm_mask),
                                ptr(pnt_mask), B, R, K, ptr(ov), ptr(st), stream_ptr()), 'gvd_iou_targets')
    return ov, st


def step_targets(overlaps, mask_boxes, frm_mask, pnt_mask, Lc):
    B, R, K = overlaps.shape
    mask_boxes = mask_boxes.contiguous()
    Lp1 = mask_boxes.shape[-1]
    roi = torch.empty(B, Lc, R, device=overlaps.device, dtype=torch.float32)
    fm = torch.empty(B, Lc, R + 1, device=overlaps.device, dtype=torch.uint8)
    check(lib().gvd_step_targets(ptr(overlaps), ptr(mask_boxes), ptr(frm_mask.contiguous()),
                                 ptr(pnt_mask.contiguous()), B, R, K, Lp1, Lc, ptr(roi), ptr(fm), stream_ptr()),
          'gvd_step_targets')
    return roi, fm


def masked_lsm_loss(x, label):
    """-mean(log_softmax(x, -1)[label != 0]) (utils.py:139,142); x, label: [..., N]."""
    require_cuda_f32(x, label)
    N = x.shape[-1]
    x2, l2 = x.reshape(-1, N), label.reshape(-1, N)
    assert x2.stride(-1) == 1 and l2.stride(-1) == 1
    acc = torch.empty(2 + 2 * x2.shape[0], device=x.device, dtype=torch.float32)
    lse = torch.empty(x2.shape[0], device=x.device, dtype=torch.float32)
    check(lib().gvd_masked_lsm_loss(ptr(x2), x2.stride(0), ptr(l2), l2.stride(0), x2.shape[0], N, ptr(acc),
                                    ptr(lse), stream_ptr()), 'gvd_masked_lsm_loss')
    masked_lsm_loss.last_acc = acc
    return acc[0] / acc[1], lse


# --------------------------------------------------------------------------------------------------
# autograd-aware entry points (forward = HIP kernel; backward GEMMs are plain library GEMMs via torch)
# --------------------------------------------------------------------------------------------------
def draw_seed():
    """A 62-bit Philox key for one dropout site and step, from torch's CPU generator (reproducible under
    torch.manual_seed; no device synchronisation).  Under torch.distributed the rank is mixed in, so that replicas seeded
    alike still draw different masks (nn.DataParallel's replicas draw from their own device generators, main.py:655)."""
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        seed ^= ((torch.distributed.get_rank() + 1) * 0x9E3779B97F4A7C15) & (2 ** 62 - 1)
    return seed


def dropout_(x, p_drop, seed):
    """In-place F.dropout (training mode) on a contiguous fp32 tensor: x <- x * keep / (1 - p), keep from Philox keyed by
    `seed` (csrc/train_fused.hip).  No mask tensor is produced: backward passes regenerate it or do without."""
    require_cuda_f32(x)
    assert x.is_contiguous() and x.numel() % 4 == 0
    check(lib().gvd_dropout_rows(ptr(x), ptr(x), x.numel(), float(p_drop), seed, stream_ptr()), 'gvd_dropout_rows')
    return x


def dropout_rows(x, p_drop, seed):
    """Out-of-place form of dropout_ (tests; the mask of the fused residual LayerNorm for the same seed)."""
    require_cuda_f32(x)
    x = x.contiguous()
    y = torch.empty_like(x)
    check(lib().gvd_dropout_rows(ptr(x), ptr(y), x.numel(), float(p_drop), seed, stream_ptr()), 'gvd_dropout_rows')
    return y


class _DropoutFn(torch.autograd.Function):
    """F.dropout (training mode) on the Philox row kernel, forward and backward: dropout is linear in x, so the backward is the
    same kernel on dy with the same seed (the mask is regenerated, never stored)."""

    @staticmethod
    def forward(ctx, x, p_drop, seed):
        ctx.cfg = (p_drop, seed)
        return dropout_rows(x, p_drop, seed)

    @staticmethod
    def backward(ctx, dy):
        p_drop, seed = ctx.cfg
        return dropout_rows(dy, p_drop, seed), None, None


def dropout(x, p_drop, training=True):
    """F.dropout(x, p_drop, training) for GPU fp32 tensors whose element count is a multiple of 4 (every dropout site of the
    hot path: token / visual-word embeddings, h_lang - model.py:79-82,428; AttModel.py:161)."""
    p_drop = float(p_drop)
    if not training or p_drop <= 0.0:
        return x
    seed = draw_seed()
    if torch.is_grad_enabled() and x.requires_grad:
        return _DropoutFn.apply(x, p_drop, seed)
    return dropout_rows(x.detach(), p_drop, seed)


class _BnReluTrainFn(torch.autograd.Function):
    """nn.BatchNorm1d(C) in TRAIN mode + ReLU over x [rows, C] (model.py:114,397 `att_embed_aux` on the [B Ft, C] layout the
    frame projections write): batch statistics + running-statistics update + normalise + ReLU in gvd_bn_train_fwd, the whole
    backward in gvd_bn_train_bwd."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum):
        rows, Cc = x.shape
        stat = torch.empty(4 * Cc, device=x.device, dtype=torch.float32)
        parts = torch.empty(lib().gvd_bn_parts(rows) * 2 * Cc, device=x.device, dtype=torch.float32)
        y = torch.empty_like(x)
        check(lib().gvd_bn_train_fwd(ptr(x), rows, Cc, ptr(weight), ptr(bias), eps, momentum, ptr(running_mean),
                                     ptr(running_var), ptr(stat), ptr(parts), ptr(y), stream_ptr()), 'gvd_bn_train_fwd')
        ctx.save_for_backward(x, y, stat)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, stat = ctx.saved_tensors
        rows, Cc = x.shape
        dy = dy.contiguous()
        parts = torch.empty(lib().gvd_bn_parts(rows) * 2 * Cc, device=x.device, dtype=torch.float32)
        sums = torch.empty(2 * Cc, device=x.device, dtype=torch.float32)
        dx = torch.empty_like(x)
        check(lib().gvd_bn_train_bwd(ptr(x), ptr(y), ptr(dy), ptr(stat), rows, Cc, ptr(parts), ptr(sums), ptr(dx),
                                     stream_ptr()), 'gvd_bn_train_bwd')
        return dx, sums[Cc:], sums[:Cc], None, None, None, None


def bn_relu_train(x, bn):
    """relu(bn(x)) for an nn.BatchNorm1d `bn` in training mode, x [rows, C] contiguous (C % 4 == 0): batch statistics,
    running statistics and num_batches_tracked updated like the module's forward does."""
    require_cuda_f32(x)
    assert x.is_contiguous() and x.dim() == 2 and bn.momentum is not None and bn.affine
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return _BnReluTrainFn.apply(x, bn.weight, bn.bias, rm, rv, float(bn.eps), float(bn.momentum))


def relu_dropout_bwd(dy, y, p_drop):
    """Backward of y = dropout(relu(z)) from y alone (y > 0 <=> z > 0 and kept): returns dz [M,N] and the bias gradient
    sum_m dz [N] from ONE pass over (dy, y) (gvd_relu_dropout_bwd_colsum; the per-workgroup partials are added in order)."""
    M, N = dy.shape
    dz = torch.empty_like(dy)
    parts = torch.empty(lib().gvd_relu_dropout_bwd_parts(M), N, device=dy.device, dtype=torch.float32)
    check(lib().gvd_relu_dropout_bwd_colsum(ptr(dy), ptr(y), ptr(dz), ptr(parts), M, N, float(p_drop), stream_ptr()),
          'gvd_relu_dropout_bwd_colsum')
    return dz, parts.sum(0)




class _LinearFn(torch.autograd.Function):
    """nn.Linear (+ReLU (+dropout)) forward on the MFMA GEMM; backward = [one fused pass: ReLU / dropout mask + bias
    gradient] + the K-strided dX / dW products."""

    @staticmethod
    def forward(ctx, x, w, b, act, p_drop, seed):
        out = gemm_nt(x, w, b, act)
        if p_drop > 0:
            dropout_(out, p_drop, seed)           # in place on the product's fresh output: y = dropout(relu(z))
        ctx.act = act
        ctx.p_drop = p_drop
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, w, out if act else None)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w, out = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        db = None
        fused = ctx.act and dy2.shape[-1] % 4 == 0 and dy2.data_ptr() % 16 == 0
        if fused:
            dy2, db = relu_dropout_bwd(dy2.contiguous(), out.reshape(-1, out.shape[-1]), ctx.p_drop)
        else:
            if ctx.p_drop > 0:
                dy2 = dy2 * (1.0 / (1.0 - ctx.p_drop))      # (out == 0 where dropped: the ReLU mask below covers the keep mask)
            if ctx.act:
                dy2 = torch.ops.aten.threshold_backward(dy2, out.reshape(-1, out.shape[-1]), 0)
            dy2 = dy2.contiguous()
        x2 = x.reshape(-1, x.shape[-1])
        dx = dw = None
        if ctx.needs_input_grad[0]:
            # MFMA kernel with W in place as the K-strided operand; few output tiles: the grouped small-M kernel
            dx = dx_any(dy2, w.detach(), what='linear dX').view_as(x)
        if ctx.needs_input_grad[1]:
            dw = dw_any(dy2, x2.detach(), what='linear dW')
        if not (ctx.has_bias and ctx.needs_input_grad[2]):
            db = None
        elif db is None:
            db = dy2.sum(0)
        return dx, dw, db, None, None, None


def linear(x, w, b=None, act=0, p_drop=0.0):
    """nn.Linear(+ReLU)(+F.dropout(p_drop), training mode): MFMA GEMM forward; differentiable when grad mode is on.
    p_drop > 0 requires act = 1 (every dropout of the hot path that follows a Linear follows its ReLU: model.py:312,363,
    384,393-395) - the backward then needs no mask."""
    p_drop = float(p_drop)
    assert p_drop == 0.0 or act == 1
    seed = draw_seed() if p_drop > 0 else 0
    if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or (b is not None and b.requires_grad)):
        return _LinearFn.apply(x, w, b, act, p_drop, seed)
    out = gemm_nt(x.detach(), w.detach(), None if b is None else b.detach(), act)
    return dropout_(out, p_drop, seed) if p_drop > 0 else out


def _stream_grounder_ok(xt, feats, xt_shared):
    """The per-caption form (any number of words: above 32 the streaming kernels run once per 32-word chunk)."""
    return (not xt_shared and xt.dim() == 3 and feats.shape[2] % 128 == 0 and xt.is_contiguous() and feats.is_contiguous())


def _m_chunks(M):
    return [(m0, min(M, m0 + 32)) for m0 in range(0, M, 32)]


def _mchunk(t, m0, m1, dim3_only=False):
    """Rows m0..m1 of a per-word operand ([B,M,.] or [B,M]); a [B,R] mask (same for every word) passes through."""
    if t is None or (dim3_only and t.dim() == 2):
        return t
    return t[:, m0:m1]


class _GrounderFn(torch.autograd.Function):
    """`_grounder` dot branch.  The per-caption form (xt [B, M <= 32, K]: model.py:469-480) runs on the three streaming
    kernels of csrc/stream_mm.hip - forward and both gradients read / write the [B,R,K] region tensor exactly once, nothing
    goes through a library GEMM; the class-similarity form (xt shared by the batch, M = D1: model.py:321-340 on configurations
    the fused row kernels do not take) keeps the batched MFMA GEMM."""

    @staticmethod
    def forward(ctx, xt, feats, mask, mbias, rowbias, xt_shared):
        ctx.stream = _stream_grounder_ok(xt, feats, xt_shared) and (mbias is None or mbias.dim() == 2)
        if ctx.stream:
            out = grounder_stream_any(xt, feats, mask, mbias, rowbias)
        else:
            out = grounder_dot(xt, feats, mask, mbias, rowbias, xt_shared)
        ctx.xt_shared = xt_shared
        ctx.save_for_backward(xt, feats, mask)
        ctx.mbias_dim = None if mbias is None else mbias.dim()
        return out

    @staticmethod
    def backward(ctx, dout):
        xt, feats, mask = ctx.saved_tensors
        g = [None] * 6
        if ctx.stream:
            B, M, R = dout.shape
            need_sum = ctx.mbias_dim is not None and ctx.needs_input_grad[3]
            # the masked gradient IS an output (the gradient of `rowbias` = the region-attention logits): one pass writes it and
            # its row sums (the gradient of the class bias); both products below read it
            if M <= 32:
                dm, rs, dmt = masked_copy_rowsum(dout, mask, want_sum=need_sum, want_t=True)
                if ctx.needs_input_grad[0]:
                    g[0] = rows_contract(dm, feats, S_t=dmt)
                if ctx.needs_input_grad[1]:
                    g[1] = rank_update(dm, xt)
            else:
                # captions above 32 words (`--seq_length 40`): the same kernels per 32-word chunk
                dms, rss, g0 = [], [], []
                for m0, m1 in _m_chunks(M):
                    dm_c, rs_c, dmt_c = masked_copy_rowsum(dout[:, m0:m1], _mchunk(mask, m0, m1, True), want_sum=need_sum,
                                                           want_t=True)
                    dms.append(dm_c); rss.append(rs_c)
                    if ctx.needs_input_grad[0]:
                        g0.append(rows_contract(dm_c, feats, S_t=dmt_c))
                    if ctx.needs_input_grad[1]:
                        part = rank_update(dm_c, xt[:, m0:m1])
                        g[1] = part if g[1] is None else g[1].add_(part)
                dm = torch.cat(dms, 1)
                rs = torch.cat(rss, 1) if need_sum else None
                if ctx.needs_input_grad[0]:
                    g[0] = torch.cat(g0, 1)
            if need_sum:
                g[3] = rs
            if ctx.needs_input_grad[4]:
                g[4] = dm
            return tuple(g)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            library_fallback('grounder backward', 'xt %s, feats %s' % (tuple(xt.shape), tuple(feats.shape)))
        if mask is not None:
            m = mask.bool()
            if m.dim() == 2:
                m = m.unsqueeze(1)
            dout = dout.masked_fill(m, 0.0)
        if ctx.needs_input_grad[0]:
            dxt = torch.matmul(dout, feats)                    # [B,M,K]
            g[0] = dxt.sum(0) if ctx.xt_shared else dxt
        if ctx.needs_input_grad[1]:
            g[1] = torch.matmul(dout.transpose(1, 2), xt)     # [B,R,K] (xt broadcasts when shared)
        if ctx.mbias_dim is not None and ctx.needs_input_grad[3]:
            sm = dout.sum(-1)
            g[3] = sm.sum(0) if ctx.mbias_dim == 1 else sm
        if ctx.needs_input_grad[4]:
            g[4] = dout
        return tuple(g)


def masked_copy_rowsum(x, mask, want_sum=True, want_t=False):
    """y = x with the entries under `mask` (u8 [B,R] or [B,M,R]) set to 0, (optionally) y.sum(-1) and (optionally, M <= 32) the
    transposed copy y_t [B,R,32] (columns m >= M zero): one pass (gvd_masked_copy_rowsum).  x [B,M,R]."""
    require_cuda_f32(x)
    B, M, R = x.shape
    x = x if x.stride(2) == 1 else x.contiguous()
    y = torch.empty(B, M, R, device=x.device, dtype=torch.float32)
    rs = torch.empty(B, M, device=x.device, dtype=torch.float32) if want_sum else None
    yt = torch.empty(B, R, 32, device=x.device, dtype=torch.float32) if want_t else None
    mp, mld, mbs = _mask_strides(mask, M)
    check(lib().gvd_masked_copy_rowsum(ptr(x), x.stride(1), x.stride(0), mp, mld, mbs, B, M, R, ptr(y), ptr(rs), ptr(yt),
                                       stream_ptr()), 'gvd_masked_copy_rowsum')
    return (y, rs, yt) if want_t else (y, rs)


def grounder(xt, feats, mask, mbias=None, rowbias=None, xt_shared=False):
    """`_grounder` dot branch (model.py:243-280); differentiable when grad mode is on."""
    ts = [t for t in (xt, feats, mbias, rowbias) if t is not None]
    if torch.is_grad_enabled() and any(t.requires_grad for t in ts):
        return _GrounderFn.apply(xt, feats, mask, mbias, rowbias, xt_shared)
    d = lambda t: None if t is None else t.detach()
    if _stream_grounder_ok(xt, feats, xt_shared) and (mbias is None or mbias.dim() == 2):
        return grounder_stream_any(d(xt), d(feats), mask, d(mbias), d(rowbias))
    return grounder_dot(d(xt), d(feats), mask, d(mbias), d(rowbias), xt_shared)


class _NllGatherFn(torch.autograd.Function):
    """log_softmax(logits)[target] per row (utils.py:131-132) without materialising the log-softmax."""

    @staticmethod
    def forward(ctx, logits, target):
        assert logits.stride(1) == 1 and target.is_contiguous()
        lse, picked, _, _ = logsoftmax_rows(logits, target)
        ctx.save_for_backward(logits, target, lse)
        return picked

    @staticmethod
    def backward(ctx, dpicked):
        logits, target, lse = ctx.saved_tensors
        rows, V = logits.shape
        g = torch.empty(rows, V, device=logits.device, dtype=torch.float32)
        dp = dpicked.contiguous()
        check(lib().gvd_nll_gather_bwd(ptr(logits), logits.stride(0), rows, V, ptr(target), ptr(lse), ptr(dp), ptr(g), V,
                                       stream_ptr()), 'gvd_nll_gather_bwd')
        return g, None


def nll_gather(logits, target):
    if torch.is_grad_enabled() and logits.requires_grad:
        return _NllGatherFn.apply(logits, target)
    return logsoftmax_rows(logits.detach(), target)[1]


class _MaskedLsmFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, label):
        loss, lse = masked_lsm_loss(x, label)
        ctx.save_for_backward(x, label, lse, masked_lsm_loss.last_acc)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        x, label, lse, acc = ctx.saved_tensors
        N = x.shape[-1]
        x2, l2 = x.reshape(-1, N), label.reshape(-1, N)
        g = torch.empty(x2.shape[0], N, device=x.device, dtype=torch.float32)
        dl = dloss.reshape(1).contiguous()
        check(lib().gvd_masked_lsm_bwd(ptr(x2), x2.stride(0), ptr(l2), l2.stride(0), x2.shape[0], N, ptr(acc), ptr(lse),
                                       ptr(dl), ptr(g), N, stream_ptr()), 'gvd_masked_lsm_bwd')
        return g.view(x.shape), None


def masked_lsm(x, label):
    """-mean(log_softmax(x,-1)[label != 0]); differentiable w.r.t. x when grad mode is on."""
    if torch.is_grad_enabled() and x.requires_grad:
        return _MaskedLsmFn.apply(x, label)
    return masked_lsm_loss(x.detach(), label)[0]


class _ClsLossFn(torch.autograd.Function):
    """-mean(clamp(log sim_mat[b, target, r], -100)) over target > 0 (model.py:345-350): fused gather + log + masked mean."""

    @staticmethod
    def forward(ctx, sim_mat, sim_target):
        B, D1, R = sim_mat.shape
        K = sim_target.shape[1]
        nblk = (B * K * R + 255) // 256
        acc = torch.empty(2 + 2 * nblk, device=sim_mat.device, dtype=torch.float32)
        check(lib().gvd_cls_loss(ptr(sim_mat), sim_mat.stride(0), sim_mat.stride(1), sim_mat.stride(2), ptr(sim_target),
                                 B, D1, R, K, ptr(acc), stream_ptr()), 'gvd_cls_loss')
        ctx.save_for_backward(sim_mat, sim_target, acc)
        return acc[0] / acc[1]

    @staticmethod
    def backward(ctx, dloss):
        sim_mat, tgt, acc = ctx.saved_tensors
        p = torch.gather(sim_mat, 1, tgt)
        g = torch.where((tgt > 0) & (p > 3.7200759760208555e-44), -(dloss / acc[1]) / p, torch.zeros_like(p))
        if sim_mat.stride(1) == 1:      # class-last in memory (training path): build the gradient in that layout
            B, D1, R = sim_mat.shape
            gt = torch.zeros(B, R, D1, device=sim_mat.device, dtype=sim_mat.dtype)
            gt.scatter_add_(2, tgt.transpose(1, 2), g.transpose(1, 2))
            return gt.transpose(1, 2), None
        return torch.zeros_like(sim_mat).scatter_add_(1, tgt, g), None


def cls_loss(sim_mat, sim_target):
    """Region-classification loss on the HIP kernel (NaN for an empty selection, like the reference's mean over nothing).
    sim_mat [B,D1,R] in any memory layout (the training path hands in the transposed view of the class-last tensor)."""
    require_cuda_f32(sim_mat)
    assert sim_mat.dim() == 3 and sim_target.is_contiguous() and sim_target.dtype == torch.int64
    if torch.is_grad_enabled() and sim_mat.requires_grad:
        return _ClsLossFn.apply(sim_mat, sim_target)
    return _ClsLossFn.forward(_NoCtx(), sim_mat.detach(), sim_target)


class _NoCtx:
    def save_for_backward(self, *a):
        pass


# --------------------------------------------------------------------------------------------------
# backward kernels of the decoder loop (used by decoder_bwd.DecoderLoopFn)
# --------------------------------------------------------------------------------------------------
def lstm_cell_bwd(dh, dc_next, gates, c_prev, c_new, dg_out=None, dh2=None):
    """Pointwise LSTMCell backward -> (d pre-activation gates [B,4H], dc_prev [B,H]); `dg_out` ([B,4H], unit inner
    stride) receives the gate gradients in place when given.  dh2: optional second addend of the hidden-state gradient
    (the recurrent contribution of the next step), added inside the kernel."""
    require_cuda_f32(dh, dh2, dc_next, gates, c_prev, c_new)
    B, H = c_prev.shape
    dh = dh if dh.stride(-1) == 1 else dh.contiguous()
    if dh2 is not None and dh2.stride(-1) != 1:
        dh2 = dh2.contiguous()
    dg = torch.empty(B, 4 * H, device=dh.device, dtype=torch.float32) if dg_out is None else dg_out
    assert dg.stride(-1) == 1 and dg.shape == (B, 4 * H)
    dcp = torch.empty(B, H, device=dh.device, dtype=torch.float32)
    check(lib().gvd_lstm_cell_bwd(ptr(dh), dh.stride(0), ptr(dh2), dh2.stride(0) if dh2 is not None else 0,
                                  ptr(dc_next), dc_next.stride(0) if dc_next is not None else 0,
                                  ptr(gates), gates.stride(0), ptr(c_prev), c_prev.stride(0), ptr(c_new),
                                  c_new.stride(0), B, H, ptr(dg), dg.stride(0), ptr(dcp), H, stream_ptr()),
          'gvd_lstm_cell_bwd')
    return dg, dcp


def attn_bwd_chunks(N, B):
    """Number of per-chunk partial slabs gvd_attn_bwd_step writes for N rows and B samples."""
    return lib().gvd_attn_bwd_chunks(N, B)


def attn_bwd_step(side, alpha, ctx, d_ctx, d_logits=None, de_out=None, dq_out=None, dw_part=None, dab_part=None,
                  dq_part=None):
    """One attention side, one step.  side: dict like attention_step's (feats, p_feats, q, w, alpha_bias
    [, att_mask, pnt_mask]).  Returns de [B,N], d_q [B,A], d_w [B,A] (per-sample partial of the alpha_net weight
    gradient), d_alpha_bias [B].
    In-place form for the BPTT loop: `de_out` [B,N] contiguous receives de, `dq_out` [B,A] (any strides) the summed
    query gradient, and `dw_part` [B,NC,A] / `dab_part` [B,NC] (NC = attn_bwd_chunks(N, B), contiguous) the UNSUMMED
    per-chunk partials — the caller reduces them once for all steps; they are then returned as such.  `dq_part`
    [B,NC,A]: the query gradient is left as per-chunk partials too (the caller sums both sides with sum_chunks_pair)."""
    f = side['feats']
    B, N, H = f.shape
    A = side['p_feats'].shape[-1]
    require_cuda_f32(f, alpha, ctx, d_ctx, d_logits)
    s = _side(**{k: v for k, v in side.items() if k not in ('logits_out', 'scores_out')})
    d_ctx = d_ctx if (d_ctx.stride(-1) == 1 and d_ctx.data_ptr() % 16 == 0 and d_ctx.stride(0) % 4 == 0) else d_ctx.contiguous()
    if d_logits is not None and d_logits.stride(-1) != 1:
        d_logits = d_logits.contiguous()
    assert alpha.stride(-1) == 1 and ctx.stride(-1) == 1
    nc = lib().gvd_attn_bwd_chunks(N, B)
    de = torch.empty(B, N, device=f.device, dtype=torch.float32) if de_out is None else de_out
    dq = torch.empty(B, nc, A, device=f.device, dtype=torch.float32) if dq_part is None else dq_part
    dw = torch.empty(B, nc, A, device=f.device, dtype=torch.float32) if dw_part is None else dw_part
    dab = torch.empty(B, nc, device=f.device, dtype=torch.float32) if dab_part is None else dab_part
    assert de.is_contiguous() and dw.is_contiguous() and dab.is_contiguous() and dq.is_contiguous()
    assert de.shape == (B, N) and dw.shape == (B, nc, A) and dab.shape == (B, nc) and dq.shape == (B, nc, A)
    check(lib().gvd_attn_bwd_step(C.byref(s), B, A, H, ptr(alpha), alpha.stride(0), ptr(ctx), ctx.stride(0),
                                  ptr(d_ctx), d_ctx.stride(0), ptr(d_logits),
                                  d_logits.stride(0) if d_logits is not None else 0, ptr(de), N, ptr(dq), ptr(dw),
                                  ptr(dab), stream_ptr()), 'gvd_attn_bwd_step')
    if dq_part is not None:
        dq_sum = dq
    else:
        dq_sum = torch.sum(dq, 1, out=dq_out) if dq_out is not None else dq.sum(1)
    return de, dq_sum, (dw.sum(1) if dw_part is None else dw), (dab.sum(1) if dab_part is None else dab)


def sum_chunks_pair(a, r, out):
    """out[:, :A] = a.sum(1), out[:, A:] = r.sum(1) for the per-chunk partials a [B,nca,A], r [B,ncr,A] of the two
    attention sides of one BPTT step (chunks added in order); out: [B, 2A] view with unit inner stride."""
    require_cuda_f32(a, r, out)
    B, nca, A = a.shape
    assert a.is_contiguous() and r.is_contiguous() and r.shape[0] == B and r.shape[2] == A
    assert out.shape == (B, 2 * A) and out.stride(-1) == 1
    check(lib().gvd_sum_chunks_pair(ptr(a), nca, ptr(r), r.shape[1], B, A, ptr(out), out.stride(0), stream_ptr()),
          'gvd_sum_chunks_pair')
    return out


def attn_bwd_pfeats(p_feats, q_all, de_all, w, score_mode=0):
    """d_p_feats [B,N,A] = sum_t de_all[t,b,n] * w * (1 - tanh^2(p_feats[b,n] + q_all[t,b])).
    q_all: [Lc,B,A] view (inner stride 1); de_all: [Lc,B,N] contiguous.  score_mode 1 ('mix_mul'): ... (1 - tanh^2(p q)) q;
    2 ('dp'): sum_t de q (w None)."""
    require_cuda_f32(p_feats, q_all, de_all, *([] if w is None else [w]))
    B, N, A = p_feats.shape
    Lc = q_all.shape[0]
    assert q_all.stride(-1) == 1 and de_all.is_contiguous() and p_feats.is_contiguous()
    out = torch.empty_like(p_feats)
    check(lib().gvd_attn_bwd_pfeats(ptr(p_feats), B, N, A, ptr(q_all), q_all.stride(0), q_all.stride(1), ptr(de_all),
                                    de_all.stride(0), de_all.stride(1), ptr(w), Lc, ptr(out), score_mode, stream_ptr()),
          'gvd_attn_bwd_pfeats')
    return out


GRU_BARRIER = os.environ.get('GVD_GRU_BARRIER', 'counter')   # 'counter' (hand-rolled) | 'cg' (library grid sync)

# Process-wide switch of the persistent kernels with the hand-rolled grid barrier (greedy decoder batch_size <= 4, bi-GRU
# recurrence).  They need all their workgroups co-resident; on a GPU shared with other work a barrier spin can run out
# (status word raised, results invalid).  The first such timeout turns them off for the rest of the process - a property
# of the device the process runs on, not of one model: the decoder then runs its kernel-per-op loop, the GRU its
# cooperative launch with the library grid sync - and the caller recomputes (att_model.TopDownModel.forward, train.Trainer).
_persistent = {'on': True, 'timeouts': 0}


def persistent_kernels_enabled():
    return _persistent['on']


def disable_persistent_kernels(n_timeouts=1):
    _persistent['timeouts'] += int(n_timeouts)
    if _persistent['on']:
        _persistent['on'] = False
        import warnings
        warnings.warn('libgvd_hip: %d persistent-kernel launch(es) hit a grid-barrier timeout (workgroups not co-resident: '
                      'a shared GPU?); recomputing and continuing on the kernel-per-op decoder / the cooperative GRU launch '
                      'for the rest of this process' % n_timeouts, RuntimeWarning, stacklevel=3)


def _spin_limit_env():
    v = os.environ.get('GVD_SPIN_LIMIT')          # test aid: forced barrier timeouts (csrc/gvd_common.h GVD_SYNC_LIMIT)
    return int(v) if v and int(v) > 0 else 0


def gru_layer(gi, w_f, b_f, w_b, b_b, B, T, Hh, flags=None, barrier=None):
    """The recurrence of one bidirectional GRU layer as ONE persistent cooperative kernel (gvd_gru_bidir_layer).
    gi [B*T, 2*3*Hh] input projections of both directions (incl. b_ih) -> out [B,T,2*Hh].  `flags` (list) receives the
    launch's barrier-timeout words (TopDownModel.check_kernel_status)."""
    require_cuda_f32(gi, w_f, b_f, w_b, b_b)
    assert gi.is_contiguous() and w_f.is_contiguous() and w_b.is_contiguous() and b_f.is_contiguous() and b_b.is_contiguous()
    out = torch.empty(B, T, 2 * Hh, device=gi.device, dtype=torch.float32)
    sync = None
    if barrier is None:
        barrier = GRU_BARRIER if _persistent['on'] else 'cg'
    if barrier == 'counter':
        sync = torch.zeros(lib().gvd_grid_sync_words() * ((B + 255) // 256), dtype=torch.int32, device=gi.device)
        if _spin_limit_env():
            sync.view(-1, lib().gvd_grid_sync_words())[:, 33] = _spin_limit_env()
    check(lib().gvd_gru_bidir_layer(ptr(gi), ptr(w_f), ptr(b_f), ptr(w_b), ptr(b_b), ptr(out), B, T, Hh,
                                    ptr(sync), stream_ptr()), 'gvd_gru_bidir_layer')
    gru_layer.last_sync = sync
    if flags is not None and sync is not None:      # word 32 of every barrier object latches a spin timeout
        flags.append(sync.view(-1, lib().gvd_grid_sync_words())[:, 32])
    return out


def lstm_seq_layer(gi, w_f, b_f, w_b, b_b, B, T, Hh, flags=None, save=False):
    """The recurrence of one bidirectional LSTM layer as ONE persistent kernel (gvd_lstm_bidir_layer; `--t_attn_mode bilstm`,
    model.py:145-149).  gi [B*T, 2*4*Hh] input projections of both directions (incl. b_ih) -> out [B,T,2*Hh]; save=True also
    returns the post-activation gates [B,T,2,4*Hh] and the cell states [B,T,2,Hh] of every step (training).  `flags` receives
    the launch's barrier-timeout words like gru_layer."""
    require_cuda_f32(gi, w_f, b_f, w_b, b_b)
    assert gi.is_contiguous() and w_f.is_contiguous() and w_b.is_contiguous() and b_f.is_contiguous() and b_b.is_contiguous()
    dev = gi.device
    out = torch.empty(B, T, 2 * Hh, device=dev, dtype=torch.float32)
    gates = torch.empty(B, T, 2, 4 * Hh, device=dev, dtype=torch.float32) if save else None
    c_seq = torch.empty(B, T, 2, Hh, device=dev, dtype=torch.float32) if save else None
    if not _persistent['on']:
        # the process switched its persistent kernels off (a grid barrier timed out on a shared GPU): one fused LSTM-cell
        # launch per (step, direction) - the decoder's cell kernel with the step's input projection as its row bias
        gi4 = gi.view(B, T, 2, 4 * Hh)
        zero = torch.zeros(B, Hh, device=dev, dtype=torch.float32)
        for d, (w, b) in enumerate(((w_f, b_f), (w_b, b_b))):
            h_prev, c_prev = zero, zero
            cbuf = [torch.empty(B, Hh, device=dev, dtype=torch.float32) for _ in range(2)]
            for i in range(T):
                t = i if d == 0 else T - 1 - i
                h_prev, c_prev = lstm_cell([], [], h_prev, w, None, b, c_prev, rowbias=gi4[:, t, d],
                                           gates_out=gates[:, t, d] if save else None, h_out=out[:, t, d * Hh:(d + 1) * Hh],
                                           c_out=c_seq[:, t, d] if save else cbuf[i & 1])
        lstm_seq_layer.last_sync = None
        return (out, gates, c_seq) if save else out
    c_state = torch.empty(B, 2, Hh, device=dev, dtype=torch.float32)
    sync = torch.zeros(lib().gvd_grid_sync_words() * ((B + 255) // 256), dtype=torch.int32, device=dev)
    if _spin_limit_env():
        sync.view(-1, lib().gvd_grid_sync_words())[:, 33] = _spin_limit_env()
    check(lib().gvd_lstm_bidir_layer(ptr(gi), ptr(w_f), ptr(b_f), ptr(w_b), ptr(b_b), ptr(out), ptr(c_state), ptr(gates),
                                     ptr(c_seq), B, T, Hh, ptr(sync), stream_ptr()), 'gvd_lstm_bidir_layer')
    lstm_seq_layer.last_sync = sync
    if flags is not None:
        flags.append(sync.view(-1, lib().gvd_grid_sync_words())[:, 32])
    return (out, gates, c_seq) if save else out


def lstm_bidir_2layer(x, lstm, flags=None, packed=None):
    """Inference forward of the frame encoder nn.LSTM(1024, 512, 2, bidirectional, batch_first) (model.py:145-149,399): per
    layer one MFMA GEMM for both directions' input projections + one persistent kernel for the recurrence.
    x [B,T,1024] -> [B,T,1024].  packed: see gru_bidir_2layer."""
    require_cuda_f32(x)
    B, T, _ = x.shape
    Hh = lstm.hidden_size
    inp = x.contiguous()
    for l in range(lstm.num_layers):
        g = lambda n: getattr(lstm, '%s_l%d' % (n, l)).detach()
        gr = lambda n: getattr(lstm, '%s_l%d_reverse' % (n, l)).detach()
        stack = lambda n: (lambda: torch.cat([g(n), gr(n)], 0))
        if packed is None:
            w_ih, b_ih = stack('weight_ih')(), stack('bias_ih')()
        else:
            w_ih = packed(('lstm_w_ih', l), (g('weight_ih'), gr('weight_ih')), stack('weight_ih'))
            b_ih = packed(('lstm_b_ih', l), (g('bias_ih'), gr('bias_ih')), stack('bias_ih'))
        gi = gemm_nt(inp.view(B * T, -1), w_ih, b_ih)                      # [B*T, 8*Hh]
        inp = lstm_seq_layer(gi, g('weight_hh').contiguous(), g('bias_hh').contiguous(), gr('weight_hh').contiguous(),
                             gr('bias_hh').contiguous(), B, T, Hh, flags=flags)
    return inp


def gru_bwd_step(dout, gi, gh, out, carry_mm, carry_z, d_gi, d_gh, B, T, Hh, t_fw, t_bw, first):
    """One reverse step of a GRU layer's BPTT for both directions (gvd_gru_bwd_step); all tensors contiguous."""
    require_cuda_f32(dout, gi, gh, out, carry_mm, carry_z, d_gi, d_gh)
    assert all(t.is_contiguous() for t in (dout, gi, gh, out, carry_mm, carry_z, d_gi, d_gh))
    check(lib().gvd_gru_bwd_step(ptr(dout), ptr(gi), ptr(gh), ptr(out), ptr(carry_mm), ptr(carry_z), ptr(d_gi), ptr(d_gh),
                                 B, T, Hh, t_fw, t_bw, 1 if first else 0, stream_ptr()), 'gvd_gru_bwd_step')


def gru_bidir_2layer(x, gru, barrier=None, flags=None, packed=None):
    """Inference forward of the frame encoder nn.GRU(1024, 512, 2, bidirectional, batch_first) (model.py:399):
    per layer one MFMA GEMM for both directions' input projections + one persistent cooperative kernel for the
    recurrence (gvd_gru_bidir_layer).  x [B,T,1024] -> [B,T,1024].
    packed: the caller's cache of derived parameter copies, `packed(key, params, build)` (TopDownModel._packed): the two
    directions' stacked input-projection weights are then built once per parameter version instead of per call (two 6 MB
    concatenations per layer: 1 % of a batch_size = 4 call)."""
    require_cuda_f32(x)
    B, T, _ = x.shape
    Hh = gru.hidden_size
    inp = x.contiguous()
    syncs = []
    for l in range(gru.num_layers):
        g = lambda n: getattr(gru, '%s_l%d' % (n, l)).detach()
        gr = lambda n: getattr(gru, '%s_l%d_reverse' % (n, l)).detach()
        stack = lambda n: (lambda: torch.cat([g(n), gr(n)], 0))
        if packed is None:
            w_ih, b_ih = stack('weight_ih')(), stack('bias_ih')()
        else:
            w_ih = packed(('gru_w_ih', l), (g('weight_ih'), gr('weight_ih')), stack('weight_ih'))
            b_ih = packed(('gru_b_ih', l), (g('bias_ih'), gr('bias_ih')), stack('bias_ih'))
        gi = gemm_nt(inp.view(B * T, -1), w_ih, b_ih)                      # [B*T, 6*Hh]
        inp = gru_layer(gi, g('weight_hh').contiguous(), g('bias_hh').contiguous(), gr('weight_hh').contiguous(),
                        gr('bias_hh').contiguous(), B, T, Hh, flags=flags, barrier=barrier)
        if gru_layer.last_sync is not None:
            syncs.append(gru_layer.last_sync)
    gru_bidir_2layer.last_sync = syncs     # tests read the timeout flags (sync_timed_out) after a device sync
    return inp


def zero_masked_rows(x, mask, mask_off=0):
    """In place: zero every row x[b, r, :] whose byte mask[b, mask_off + r] is set (gvd_zero_masked_rows)."""
    require_cuda_f32(x)
    assert x.is_contiguous() and x.dim() == 3 and mask.dtype == torch.uint8 and mask.is_contiguous() and mask.is_cuda
    B, N, D = x.shape
    assert mask.shape[0] == B and mask.shape[1] >= N + mask_off
    check(lib().gvd_zero_masked_rows(ptr(x), B * N, D, ptr(mask), N, mask.shape[1], mask_off, stream_ptr()),
          'gvd_zero_masked_rows')
    return x


def sync_timed_out(sync):
    """True when any barrier object in `sync` (gvd_grid_sync_words() words each) raised its timeout word."""
    return int(sync.view(-1, lib().gvd_grid_sync_words())[:, 32].sum()) != 0


def add_layernorm_unbiased(x, y, gamma, beta, eps=1e-6, rows_dev=None):
    """gamma * (s - mean) / (std_unbiased + eps) + beta, s = x + y: residual + the encoder LayerNorm in one pass."""
    require_cuda_f32(x, y, gamma, beta)
    D = x.shape[-1]
    x2 = x.reshape(-1, D)
    y2 = y.reshape(-1, D) if y is not None else None
    assert x2.is_contiguous() and (y2 is None or y2.is_contiguous())
    out = torch.empty_like(x2)
    check(lib().gvd_add_layernorm_unbiased(ptr(x2), ptr(y2), ptr(gamma), ptr(beta), ptr(out), x2.shape[0], ptr(rows_dev),
                                           D, eps, stream_ptr()), 'gvd_add_layernorm_unbiased')
    return out.view_as(x)


class _AddLnFn(torch.autograd.Function):
    """ResidualBlock + the encoder's LayerNorm (transformer.py:66-88) as ONE row kernel forward and ONE backward (training
    path; the reference runs ~8 elementwise / reduction passes forward and ~15 backward over [B*R, 1024]).  p_drop > 0:
    the block's branch dropout (transformer.py:84,87) is applied inside both kernels (Philox mask regenerated from `seed`)."""

    @staticmethod
    def forward(ctx, x, y, gamma, beta, eps, p_drop, seed):
        if p_drop > 0:
            D = x.shape[-1]
            x2, y2 = x.reshape(-1, D), y.reshape(-1, D)
            out = torch.empty_like(x2)
            check(lib().gvd_add_layernorm_unbiased_drop(ptr(x2), ptr(y2), ptr(gamma), ptr(beta), ptr(out), x2.shape[0], D,
                                                        eps, p_drop, seed, stream_ptr()), 'gvd_add_layernorm_unbiased_drop')
            out = out.view_as(x)
        else:
            out = add_layernorm_unbiased(x, y, gamma, beta, eps)
        ctx.save_for_backward(x, y, gamma)
        ctx.cfg = (eps, p_drop, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, y, gamma = ctx.saved_tensors
        eps, p_drop, seed = ctx.cfg
        D = x.shape[-1]
        x2, y2, d2 = x.reshape(-1, D), y.reshape(-1, D), dout.reshape(-1, D).contiguous()
        rows = x2.shape[0]
        ds = torch.empty_like(x2)
        nparts = lib().gvd_add_layernorm_unbiased_bwd_parts(rows)
        parts = torch.empty(nparts, 2, D, device=x.device, dtype=torch.float32)
        if p_drop > 0:
            dy = torch.empty_like(x2)
            check(lib().gvd_add_layernorm_unbiased_drop_bwd(ptr(x2), ptr(y2), ptr(d2), ptr(gamma), ptr(ds), ptr(dy),
                                                            ptr(parts), rows, D, eps, p_drop, seed, stream_ptr()),
                  'gvd_add_layernorm_unbiased_drop_bwd')
            dy = dy.view_as(x)
        else:
            check(lib().gvd_add_layernorm_unbiased_bwd(ptr(x2), ptr(y2), ptr(d2), ptr(gamma), ptr(ds), ptr(parts), rows, D,
                                                       eps, stream_ptr()), 'gvd_add_layernorm_unbiased_bwd')
            dy = None
        ps = parts.sum(0)
        ds = ds.view_as(x)
        return ds, (ds if dy is None else dy), ps[0], ps[1], None, None, None


def add_layernorm(x, y, gamma, beta, eps=1e-6, p_drop=0.0):
    """LayerNorm_unbiased(x + dropout(y, p_drop)): differentiable (fused forward + backward row kernels) when grad mode is
    on.  p_drop is the TRAINING-mode drop probability of the branch (0: no dropout)."""
    p_drop = float(p_drop)
    seed = draw_seed() if p_drop > 0 else 0
    if torch.is_grad_enabled() and any(t.requires_grad for t in (x, y, gamma, beta)):
        return _AddLnFn.apply(x.contiguous(), y.contiguous(), gamma, beta, eps, p_drop, seed)
    if p_drop > 0:
        y = dropout_rows(y.detach(), p_drop, seed)
    return add_layernorm_unbiased(x.contiguous(), y.contiguous(), gamma.detach(), beta.detach(), eps)


def _bgemm(A, a_off, lda, a_bs, W, w_off, ldw, w_bs, K, Cout, c_off, ldc, c_bs, M, N, batch, a_t=0, w_t=0, what='bgemm',
           inner=0, a_is=0, w_is=0, c_is=0):
    """One batched launch of the MFMA GEMM on sub-blocks of larger tensors (element offsets into A / W / Cout).
    inner > 1: two-level batch of `batch` entries = outer x inner, bases outer * (a|w|c)_bs + inner * (a|w|c)_is."""
    g = GemmArgs()
    g.nseg = 1
    g.seg[0] = GemmSeg(C.c_void_p(A.data_ptr() + 4 * a_off), lda, a_bs, C.c_void_p(W.data_ptr() + 4 * w_off), ldw, w_bs, K)
    g.C = C.c_void_p(Cout.data_ptr() + 4 * c_off); g.ldc = ldc; g.c_batch_stride = c_bs
    g.M, g.N, g.batch, g.act = M, N, batch, 0
    g.a_kstrided, g.w_kstrided = a_t, w_t
    if inner > 1:
        g.batch_inner, g.a_inner_stride, g.w_inner_stride, g.c_inner_stride = inner, a_is, w_is, c_is
    check(lib().gvd_gemm_nt_f32(C.byref(g), stream_ptr()), 'gvd_gemm_nt_f32(%s)' % what)


def _heads_bgemm(nh, A, a_off, lda, a_bs, a_hs, W, w_off, ldw, w_bs, w_hs, K, Cout, c_off, ldc, c_bs, c_hs, M, N, B, a_t=0,
                 w_t=0, what='bgemm'):
    """The same product for every (sample, head): ONE launch over the two-level batch B x nh (heads live at stride *_hs
    inside the sample's block)."""
    _bgemm(A, a_off, lda, a_bs, W, w_off, ldw, w_bs, K, Cout, c_off, ldc, c_bs, M, N, B * nh, a_t=a_t, w_t=w_t, what=what,
           inner=nh, a_is=a_hs, w_is=w_hs, c_is=c_hs)


def enc_dropout_mask(n_maps, Rp, p_drop, seed, device='cuda'):
    """Test aid: the keep mask of the training attention core's dropout for `seed`, u8 [n_maps, Rp, Rp]."""
    out = torch.empty(n_maps, Rp, Rp, dtype=torch.uint8, device=device)
    check(lib().gvd_enc_dropout_mask(ptr(out), n_maps, Rp, float(p_drop), seed, stream_ptr()), 'gvd_enc_dropout_mask')
    return out


ENC_SCORES_MAX_BYTES = 16 << 30     # per layer; above it the backward multiplies Q K^T again instead of keeping the map


def enc_core_scores(B, nh, Rp, device):
    """The [B * nh, Rp, Rp] map the training forward hands to the backward (its log2-domain scaled + biased scores: the backward
    maps kernel loads them instead of multiplying Q K^T again - one MFMA product instead of two; 4 Rp^2 bytes per (sample,
    head): 1.6 GB per layer at batch_size = 64 x 1000 regions, kept from the layer's forward to its backward).  None when the
    map would exceed ENC_SCORES_MAX_BYTES (64 segments x 3000 regions): the backward then recomputes the product as before."""
    if 4 * B * nh * Rp * Rp > ENC_SCORES_MAX_BYTES:
        return None
    return torch.empty(B * nh, Rp, Rp, device=device, dtype=torch.float32)


def _enc_core_fwd(qkv, O, lse, B, Rp, R, Rs, nh, scale, p_drop, seed, key_bias, scores=None):
    """Flash-style forward of the training attention core (csrc/flash_attn_pad.hip, TRAIN form).  qkv [>= B*Rs, 3*nh*HP] /
    O [>= B*Rs, nh*HP] row arrays with Rs rows between consecutive samples; lse [B*nh, Rp]; scores: enc_core_scores(...) when
    a backward will follow."""
    W3 = qkv.shape[-1]
    check(lib().gvd_flash_attn_train_fwd_f32(ptr(qkv), W3, ptr(O), O.shape[-1], ptr(lse), ptr(scores), B, Rp, R, Rs, nh, HEAD_PAD,
                                             scale, ptr(key_bias), p_drop, seed, stream_ptr()), 'gvd_flash_attn_train_fwd_f32')


def _enc_core_bwd(qkv, O, lse, key_bias, dO, dqkv, B, Rp, R, Rs, nh, scale, p_drop, seed, scores=None):
    """Backward of the core: the maps kernel (csrc/enc_attn_bwd.hip) + the three one-head-slot products (csrc/gemm_n192.hip)
    into dqkv (rows < R of every sample; the caller zeroes pad rows where the layout has them).  The K-strided operands
    (dO, qkv) are read Rp rows deep per sample: with Rs < Rp the rows past a sample's last belong to the next sample (or to
    the caller's slack after the last one) and meet exact zeros of the maps."""
    dev = qkv.device
    HP, W3 = HEAD_PAD, qkv.shape[-1]
    ko, vo = nh * HP, 2 * nh * HP
    delta = torch.empty(B * nh, Rp, device=dev, dtype=torch.float32)
    Pd = torch.empty(B, nh, Rp, Rp, device=dev, dtype=torch.float32)
    dS = torch.empty(B, nh, Rp, Rp, device=dev, dtype=torch.float32)
    check(lib().gvd_enc_attn_bwd_maps(ptr(qkv), W3, ptr(dO), ptr(O), nh * HP, ptr(lse), ptr(key_bias), ptr(scores), ptr(delta),
                                      ptr(Pd), ptr(dS), B, Rp, R, Rs, nh, HP, scale, p_drop, seed, stream_ptr()),
          'gvd_enc_attn_bwd_maps')
    mb, ms = nh * Rp * Rp, Rp * Rp
    # dV_h = Pd_h^T dO_h   (both operands K-strided; pad rows of Pd are zero)
    _heads_bgemm(nh, Pd, 0, Rp, mb, ms, dO, 0, nh * HP, Rs * nh * HP, HP, Rp, dqkv, vo, W3, Rs * W3, HP, R, HP, B,
                 a_t=1, w_t=1, what='P^T dO')
    # dQ_h = dS_h K_h ;  dK_h = dS_h^T Q_h
    _heads_bgemm(nh, dS, 0, Rp, mb, ms, qkv, ko, W3, Rs * W3, HP, Rp, dqkv, 0, W3, Rs * W3, HP, R, HP, B, w_t=1,
                 what='dS K')
    _heads_bgemm(nh, dS, 0, Rp, mb, ms, qkv, 0, W3, Rs * W3, HP, Rp, dqkv, ko, W3, Rs * W3, HP, R, HP, B, a_t=1, w_t=1,
                 what='dS^T Q')


class _EncAttnCoreFn(torch.autograd.Function):
    """Self-attention core of one encoder layer on the training path (transformer.py:90-117): per head
    softmax(Q K^T / sqrt(d)) -> dropout -> @ V.

    Forward: the flash-style kernel of the inference path in its training form (csrc/flash_attn_pad.hip: dropout on the
    probabilities in registers, per-key bias, logsumexp written out) - no [B, heads, Rp, Rp] map is written or kept.
    Backward: ONE kernel recomputes the probabilities tile by tile from Q K^T and the saved logsumexp next to dO V^T and
    writes the two maps the remaining products need (csrc/enc_attn_bwd.hip: Pd for dV = Pd^T dO, dS for dQ = dS K and
    dK = dS^T Q, on the one-head-slot MFMA GEMM with K-strided operands); the maps live for the duration of this call.

    qkv: [B, Rp, 3 * nh * HP] (packed q | k | v, heads padded to HP = 176 columns, Rp % 32 == 0; the pad rows R..Rp-1 may
    hold anything finite).  Returns O [B, Rp, nh * HP] (pad rows zero).  (The stand-alone form on the padded layout: the
    training step itself runs the whole encoder layer as ONE autograd function, _EncLayerFn below, on the unpadded rows.)"""

    @staticmethod
    def forward(ctx, qkv, R, nh, scale, p_drop, seed, key_bias=None):
        require_cuda_f32(qkv, key_bias)
        assert qkv.is_contiguous()
        B, Rp, W3 = qkv.shape
        HP = W3 // (3 * nh)
        assert W3 == 3 * nh * HP and HP == HEAD_PAD and Rp % 32 == 0 and R % 4 == 0 and 4 <= R <= Rp
        dev = qkv.device
        if key_bias is not None:       # compacted layout: per-sample key weights (train_compact.py)
            assert key_bias.shape == (B, Rp) and key_bias.is_contiguous()
        # (the kernel writes all nh * HP columns of the R live rows: only the pad rows need zeros)
        O = torch.empty(B, Rp, nh * HP, device=dev, dtype=torch.float32)
        if Rp > R:
            O[:, R:].zero_()
        lse = torch.empty(B * nh, Rp, device=dev, dtype=torch.float32)
        scores = enc_core_scores(B, nh, Rp, dev) if qkv.requires_grad else None
        _enc_core_fwd(qkv, O, lse, B, Rp, R, Rp, nh, scale, p_drop, seed, key_bias, scores)
        ctx.save_for_backward(qkv, O, lse, key_bias, scores)
        ctx.cfg = (R, nh, scale, p_drop, seed)
        return O

    @staticmethod
    def backward(ctx, dO):
        qkv, O, lse, key_bias, scores = ctx.saved_tensors
        R, nh, scale, p_drop, seed = ctx.cfg
        B, Rp, W3 = qkv.shape
        dO = dO.contiguous()
        # (the dQ / dK / dV products write all 3 * nh * HP columns of the R live rows: only the pad rows need zeros)
        dqkv = torch.empty_like(qkv)
        if Rp > R:
            dqkv[:, R:].zero_()
        _enc_core_bwd(qkv, O, lse, key_bias, dO, dqkv, B, Rp, R, Rp, nh, scale, p_drop, seed, scores)
        return dqkv, None, None, None, None, None, None


def enc_attn_core(qkv, R, n_heads, scale, p_drop=0.0, key_bias=None, seed=None):
    """See _EncAttnCoreFn.  The dropout seed is drawn from torch's CPU generator (reproducible under torch.manual_seed)
    unless given.  key_bias: optional f32 [B, Rp] added to every query's scaled scores of a key (the compacted training
    layout)."""
    if seed is None:
        seed = draw_seed() if p_drop > 0 else 0
    return _EncAttnCoreFn.apply(qkv, R, n_heads, scale, float(p_drop), seed, key_bias)


def _add_ln_fwd(x, y, gamma, beta, eps, p_drop, seed):
    if p_drop > 0:
        out = torch.empty_like(x)
        check(lib().gvd_add_layernorm_unbiased_drop(ptr(x), ptr(y), ptr(gamma), ptr(beta), ptr(out), x.shape[0], x.shape[1],
                                                    eps, p_drop, seed, stream_ptr()), 'gvd_add_layernorm_unbiased_drop')
        return out
    return add_layernorm_unbiased(x, y, gamma, beta, eps)


def _add_ln_bwd(x, y, dout, gamma, eps, p_drop, seed):
    """-> ds (gradient of x), dy (gradient of the branch y: ds itself without dropout), dgamma, dbeta."""
    rows, D = x.shape
    ds = torch.empty_like(x)
    parts = torch.empty(lib().gvd_add_layernorm_unbiased_bwd_parts(rows), 2, D, device=x.device, dtype=torch.float32)
    if p_drop > 0:
        dy = torch.empty_like(x)
        check(lib().gvd_add_layernorm_unbiased_drop_bwd(ptr(x), ptr(y), ptr(dout), ptr(gamma), ptr(ds), ptr(dy), ptr(parts),
                                                        rows, D, eps, p_drop, seed, stream_ptr()),
              'gvd_add_layernorm_unbiased_drop_bwd')
    else:
        check(lib().gvd_add_layernorm_unbiased_bwd(ptr(x), ptr(y), ptr(dout), ptr(gamma), ptr(ds), ptr(parts), rows, D, eps,
                                                   stream_ptr()), 'gvd_add_layernorm_unbiased_bwd')
        dy = ds
    ps = parts.sum(0)
    return ds, dy, ps[0], ps[1]


class _EncLayerFn(torch.autograd.Function):
    """ONE encoder layer of the region encoder on the training path (transformer.py:39-133: multi-head self-attention ->
    ResidualBlock -> feed-forward -> ResidualBlock) as ONE autograd function over the region rows of the batch packed back to
    back, x [B * Rs, d]:

      forward   qkv = x W_qkv^T (heads in 176-column slots) -> flash-style core -> att = O W_o^T -> x1 = LN(x + drop(att))
                -> h = relu(x1 W_1^T + b_1) -> y = h W_2^T + b_2 -> x2 = LN(x1 + drop(y))
      backward  hand-scheduled: every dX product takes the gradient its input also receives through the residual connection
                as the ADDEND of its epilogue (dx1 = dz W_1 + ds2, dx = dqkv W_qkv + ds1 - autograd's two [B R, 1024] adds per
                layer are gone), ReLU mask + bias gradient in one pass, the packed-weight gradients sliced back to
                wq / wk / wv / wo by index.

    Rs = rows between samples: R itself when R % 4 == 0 (no pad rows anywhere: the Linear layers work on exactly the live
    rows - 2.3 % fewer flops than on the 32-row-padded layout at R = 1000 and no pad / unpad copies; only the core's maps
    and statistics keep the padded pitch Rp), else Rp with the rows R .. R4-1 masked as keys (R4 = R rounded up to 4)."""

    @staticmethod
    def forward(ctx, x, wq, wk, wv, wo, g1, be1, w1, b1, w2, b2, g2, be2, cfg):
        B, R, Rs, Rp, nh, scale, key_bias, idx, idx3, eps1, eps2, p_att, p_res1, p_res2, seeds = cfg
        HP = HEAD_PAD
        dev = x.device
        d = x.shape[1]
        rows = B * Rs
        assert x.shape == (rows, d) and x.is_contiguous()
        w_qkv = torch.zeros(3 * nh * HP, d, device=dev, dtype=torch.float32).index_copy_(0, idx3, torch.cat([wq, wk, wv], 0))
        w_o = torch.zeros(d, nh * HP, device=dev, dtype=torch.float32).index_copy_(1, idx, wo)
        slack = Rp - Rs if Rs < Rp else 0             # rows the K-strided backward products read past the last sample
        qkv_buf = torch.empty(rows + slack, 3 * nh * HP, device=dev, dtype=torch.float32)
        if slack:
            qkv_buf[rows:].zero_()
        qkv = gemm_nt(x, w_qkv, out=qkv_buf[:rows])
        O = torch.empty(rows, nh * HP, device=dev, dtype=torch.float32)
        if Rs > R:
            O.view(B, Rs, -1)[:, R:].zero_()
        lse = torch.empty(B * nh, Rp, device=dev, dtype=torch.float32)
        scores = enc_core_scores(B, nh, Rp, dev)              # (this function only runs under autograd: a backward follows)
        _enc_core_fwd(qkv_buf, O, lse, B, Rp, R, Rs, nh, scale, p_att, seeds[0], key_bias, scores)
        att = gemm_nt(O, w_o)
        x1 = _add_ln_fwd(x, att, g1, be1, eps1, p_res1, seeds[1])
        h = gemm_nt(x1, w1, b1, 1)
        y = gemm_nt(h, w2, b2)
        x2 = _add_ln_fwd(x1, y, g2, be2, eps2, p_res2, seeds[2])
        ctx.save_for_backward(x, qkv_buf, O, lse, att, x1, h, y, w_qkv, w_o, g1, w1, w2, g2, scores)
        ctx.cfg = cfg
        return x2

    @staticmethod
    def backward(ctx, dx2):
        x, qkv_buf, O, lse, att, x1, h, y, w_qkv, w_o, g1, w1, w2, g2, scores = ctx.saved_tensors
        B, R, Rs, Rp, nh, scale, key_bias, idx, idx3, eps1, eps2, p_att, p_res1, p_res2, seeds = ctx.cfg
        HP = HEAD_PAD
        dev = x.device
        rows, d = x.shape
        need = ctx.needs_input_grad
        dx2 = dx2.contiguous()
        # ---- feed-forward ResidualBlock
        ds2, dy, dg2, dbe2 = _add_ln_bwd(x1, y, dx2, g2, eps2, p_res2, seeds[2])
        db2 = dy.sum(0)
        dw2 = dw_any(dy, h, 'encoder dW (linear2)')
        dh = dx_any(dy, w2, what='encoder dX (linear2)')
        dz, db1 = relu_dropout_bwd(dh, h, 0.0)                     # ReLU mask + bias gradient in one pass
        dw1 = dw_any(dz, x1, 'encoder dW (linear1)')
        dx1 = dx_any(dz, w1, addend=ds2, what='encoder dX (linear1)')        # + the residual path's gradient
        # ---- self-attention ResidualBlock
        ds1, datt, dg1, dbe1 = _add_ln_bwd(x, att, dx1, g1, eps1, p_res1, seeds[1])
        dw_o = dw_any(datt, O, 'encoder dW (wo)')                            # [d, nh * HP]
        slack = Rp - Rs if Rs < Rp else 0
        dO_buf = torch.empty(rows + slack, nh * HP, device=dev, dtype=torch.float32)
        if slack:
            dO_buf[rows:].zero_()
        dx_any(datt, w_o, out=dO_buf[:rows], what='encoder dX (wo)')
        dqkv = torch.empty(rows, 3 * nh * HP, device=dev, dtype=torch.float32)
        if Rs > R:
            dqkv.view(B, Rs, -1)[:, R:].zero_()
        _enc_core_bwd(qkv_buf, O, lse, key_bias, dO_buf, dqkv, B, Rp, R, Rs, nh, scale, p_att, seeds[0], scores)
        dw_qkv = dw_any(dqkv, x, 'encoder dW (q|k|v)')                       # [3 * nh * HP, d]
        dx = dx_any(dqkv, w_qkv, addend=ds1, what='encoder dX (q|k|v)') if need[0] else None
        dq, dk, dv = dw_qkv.index_select(0, idx3).chunk(3, 0)
        dwo = dw_o.index_select(1, idx)
        return dx, dq, dk, dv, dwo, dg1, dbe1, dw1, db1, dw2, db2, dg2, dbe2, None


def enc_layer_ok(R, d, n_heads=6):
    """Shapes the fused training layer takes: d_model 1024 (the row kernels), heads of <= 176 columns, at most 4096 padded
    rows per sample (the key bias the flash-style core stages in LDS: 16 KB next to its three 48 KB tile buffers = the CU's
    160 KB; 40 sampled frames x 100 proposals)."""
    return d == 1024 and -(-d // n_heads) <= HEAD_PAD and -(-R // 32) * 32 <= 4096 and R >= 1


def enc_layer_rows(B, R):
    """Rows between consecutive samples of the fused training layer's row arrays: R itself (the region rows of the batch
    packed back to back, no pad rows) when the kernels take it - R % 4 == 0 (16-byte head slots of the K-strided products) and
    B * R a multiple of the K-strided weight-gradient kernel's 32-deep tile - else R rounded up to 32."""
    return R if (R % 4 == 0 and (B * R) % 32 == 0) else -(-R // 32) * 32


def enc_layer(x, lay, B, R, scale, training, key_bias=None, n_heads=6):
    """x [B * Rs, d] -> [B * Rs, d]: one encoder layer (see _EncLayerFn); `lay` = the reference's EncoderLayer parameter tree
    (att_model._build_obj_interact).  Rs = enc_layer_rows(B, R): R, or R rounded up to 32 (the caller pads; pad rows hold
    anything finite).  key_bias: optional [B, R] per-sample key weights (the compacted training layout)."""
    d = x.shape[1]
    dev = x.device
    Rp = -(-R // 32) * 32
    R4 = -(-R // 4) * 4
    Rs = enc_layer_rows(B, R)
    assert x.shape[0] == B * Rs
    sizes = [len(c) for c in torch.arange(d).chunk(n_heads)]              # torch.chunk's head widths (transformer.py:103)
    idx = torch.cat([torch.arange(sizes[h]) + h * HEAD_PAD for h in range(n_heads)]).to(dev)
    idx3 = torch.cat([idx + j * n_heads * HEAD_PAD for j in range(3)])
    kb = None
    if key_bias is not None or R4 != R:
        kb = torch.full((B, Rp), float('-inf'), device=dev, dtype=torch.float32)
        kb[:, :R] = 0.0 if key_bias is None else key_bias.float()
        if R4 != R:
            kb[:, R:R4] = -1e30                  # the rows R .. R4-1 travel as queries / keys of the kernels: no such key
    sa, ff = lay.selfattn.layer, lay.feedforward.layer
    p_att = float(sa.attention.dropout.p) if training else 0.0
    p1 = float(lay.selfattn.dropout.p) if training else 0.0
    p2 = float(lay.feedforward.dropout.p) if training else 0.0
    seeds = tuple(draw_seed() if p > 0 else 0 for p in (p_att, p1, p2))
    cfg = (B, R4, Rs, Rp, n_heads, 1.0 / scale, kb, idx, idx3, lay.selfattn.layernorm.eps, lay.feedforward.layernorm.eps,
           p_att, p1, p2, seeds)
    ln1, ln2 = lay.selfattn.layernorm, lay.feedforward.layernorm
    return _EncLayerFn.apply(x, sa.wq.weight, sa.wk.weight, sa.wv.weight, sa.wo.weight, ln1.gamma, ln1.beta,
                             ff.linear1.weight, ff.linear1.bias, ff.linear2.weight, ff.linear2.bias, ln2.gamma, ln2.beta, cfg)


def region_feature_rows(g_pool, loc, sim_logits_t, pnt_mask, ln_eps=1e-5, pad_to=1, n_cls=None):
    """[LN(g_pool) | LN(loc) | LN(softmax_classes(masked sim logits))] per proposal (model.py:336-364) in one pass.
    g_pool [B,R,2048], loc [B,R,n_loc], sim_logits_t [B,R,D1] (class-last), pnt_mask u8 [B,R+1].
    Returns pool_in [B,R,K] (K = 2048+n_loc+D1 rounded up to a multiple of `pad_to`, pad columns zero) and the class
    distribution sim_t [B,R,D1]."""
    require_cuda_f32(g_pool, loc, sim_logits_t)
    B, R, G = g_pool.shape
    n_loc, ld = loc.shape[-1], sim_logits_t.shape[-1]
    n_cls = ld if n_cls is None else n_cls          # logits rows may carry zero-padded classes past n_cls (ignored)
    assert g_pool.is_contiguous() and loc.is_contiguous() and sim_logits_t.is_contiguous()
    assert pnt_mask.dtype == torch.uint8 and pnt_mask.is_contiguous() and pnt_mask.shape == (B, R + 1)
    K = (G + n_loc + n_cls + pad_to - 1) // pad_to * pad_to
    out = torch.empty(B, R, K, device=g_pool.device, dtype=torch.float32)
    sim = torch.empty(B, R, n_cls, device=g_pool.device, dtype=torch.float32)
    mask_ptr = C.c_void_p(pnt_mask.data_ptr() + 1)              # skip the legacy pad column (main.py:227)
    check(lib().gvd_region_feature_rows(ptr(g_pool), ptr(loc), n_loc, ptr(sim_logits_t), n_cls, ld, mask_ptr, R, R + 1,
                                        ptr(out), K, ptr(sim), B * R, None, G, ln_eps, stream_ptr()),
          'gvd_region_feature_rows')
    return out, sim


class _RegionRowsFn(torch.autograd.Function):
    """Training form of `region_feature_rows`: forward = the same row kernel (logits rows `n_cls_ld` apart: the class
    axis is zero-padded to a 32-multiple so that the backward products of the similarity GEMM run on the MFMA kernel),
    backward = gvd_region_feature_rows_bwd (three layer-norm backwards + the class-softmax backward in one pass, taking
    the direct gradient of the class distribution from the region-classification loss as well)."""

    @staticmethod
    def forward(ctx, g_pool, loc, logits_pad, pnt_mask, n_cls, pad_to, ln_eps):
        B, R, G = g_pool.shape
        n_loc, ld = loc.shape[-1], logits_pad.shape[-1]
        K = (G + n_loc + n_cls + pad_to - 1) // pad_to * pad_to
        out = torch.empty(B, R, K, device=g_pool.device, dtype=torch.float32)
        sim = torch.empty(B, R, n_cls, device=g_pool.device, dtype=torch.float32)
        mask_ptr = C.c_void_p(pnt_mask.data_ptr() + 1)
        check(lib().gvd_region_feature_rows(ptr(g_pool), ptr(loc), n_loc, ptr(logits_pad), n_cls, ld, mask_ptr, R, R + 1,
                                            ptr(out), K, ptr(sim), B * R, None, G, ln_eps, stream_ptr()),
              'gvd_region_feature_rows(train)')
        ctx.save_for_backward(g_pool, loc, sim, pnt_mask)
        ctx.dims = (n_cls, ld, ln_eps)
        return out, sim

    @staticmethod
    def backward(ctx, d_out, d_sim):
        g_pool, loc, sim, pnt_mask = ctx.saved_tensors
        n_cls, ld, ln_eps = ctx.dims
        B, R, G = g_pool.shape
        n_loc = loc.shape[-1]
        d_out = d_out.contiguous()
        d_sim = None if d_sim is None else d_sim.contiguous()
        d_g = torch.empty_like(g_pool)
        d_loc = torch.empty_like(loc)
        d_logits = torch.empty(B, R, ld, device=g_pool.device, dtype=torch.float32)
        mask_ptr = C.c_void_p(pnt_mask.data_ptr() + 1)
        check(lib().gvd_region_feature_rows_bwd(ptr(g_pool), ptr(loc), n_loc, ptr(sim), n_cls, mask_ptr, R, R + 1,
                                                ptr(d_out), d_out.shape[-1], ptr(d_sim), ptr(d_g), ptr(d_loc),
                                                ptr(d_logits), ld, B * R, G, ln_eps, stream_ptr()),
              'gvd_region_feature_rows_bwd')
        return d_g, d_loc, d_logits, None, None, None, None


def region_feature_rows_train(g_pool, loc, logits_pad, pnt_mask, n_cls, pad_to=32, ln_eps=1e-5):
    """Differentiable `region_feature_rows` (see _RegionRowsFn).  logits_pad [B,R,ld >= n_cls] class-last similarity
    logits (columns >= n_cls ignored; they receive zero gradient).  Returns pool_in [B,R,K], sim_t [B,R,n_cls]."""
    require_cuda_f32(g_pool, loc, logits_pad)
    B, R, _ = g_pool.shape
    assert pnt_mask.dtype == torch.uint8 and pnt_mask.is_contiguous() and pnt_mask.shape == (B, R + 1)
    return _RegionRowsFn.apply(g_pool.contiguous(), loc.contiguous(), logits_pad.contiguous(), pnt_mask, n_cls, pad_to,
                               ln_eps)


def region_feature_rows_compact(g_pool, loc, sim_logits, row_mask, rows_dev, pad_to=32, ln_eps=1e-5, n_cls=None):
    """`region_feature_rows` over a compacted row set: g_pool [M,2048], loc [M,n_loc], sim_logits [M,ld >= n_cls] (columns
    past n_cls, the zero-padded classes of the similarity GEMM, are ignored), row_mask u8 [M] (1 = the row is a masked
    proposal), live rows = *rows_dev."""
    require_cuda_f32(g_pool, loc, sim_logits)
    M, G = g_pool.shape
    n_loc, ld = loc.shape[-1], sim_logits.shape[-1]
    n_cls = ld if n_cls is None else n_cls
    assert g_pool.is_contiguous() and loc.is_contiguous() and sim_logits.is_contiguous() and row_mask.dtype == torch.uint8
    K = (G + n_loc + n_cls + pad_to - 1) // pad_to * pad_to
    out = torch.empty(M, K, device=g_pool.device, dtype=torch.float32)
    sim = torch.empty(M, n_cls, device=g_pool.device, dtype=torch.float32)
    # mask addressing row_mask[(row / rows_per_batch) * ld + row % rows_per_batch] with one "batch" of M rows
    check(lib().gvd_region_feature_rows(ptr(g_pool), ptr(loc), n_loc, ptr(sim_logits), n_cls, ld, ptr(row_mask), M, 0,
                                        ptr(out), K, ptr(sim), M, ptr(rows_dev), G, ln_eps, stream_ptr()),
          'gvd_region_feature_rows(compact)')
    return out, sim


HEAD_PAD = 176      # padded head width of the fused obj_interact attention (11 MFMA k-blocks of 16)


def flash_attn_padded(qkv, n_heads, scale, ragged=None):
    """softmax(scale * q_h k_h^T) v_h for the heads of a fused, head-padded projection: qkv [B,R,3*n_heads*176] holds
    [q | k | v], head h of each in columns [176 h, 176 h + width) with zero pads (att_model packs the weights that way).
    Returns o [B,R,n_heads*176] in the same padded layout.
    ragged = (B, R_max, off i32 [B+1], key_w f32 [B]): qkv is [M, 3*W] with sample b in rows off[b]..off[b+1]-1, whose
    last row counts 2^key_w[b] times as a key (compact.py)."""
    require_cuda_f32(qkv)
    W = n_heads * HEAD_PAD
    assert qkv.shape[-1] == 3 * W and qkv.is_contiguous()
    if ragged is None:
        B, R, _ = qkv.shape
        o = torch.empty(B, R, W, device=qkv.device, dtype=torch.float32)
        off = kw = None
    else:
        B, R, off, kw = ragged
        o = torch.empty(qkv.shape[0], W, device=qkv.device, dtype=torch.float32)
        assert off.dtype == torch.int32 and kw.dtype == torch.float32
    base = qkv.data_ptr()
    ws = None
    if off is not None:      # ragged: device-built tile map (live workgroups first, in (sample, head, tile) order)
        ws = torch.empty(lib().gvd_flash_attn_workspace_bytes(B, R) // 4, dtype=torch.int32, device=qkv.device)
    check(lib().gvd_flash_attn_padded_f32(C.c_void_p(base), C.c_void_p(base + 4 * W), C.c_void_p(base + 8 * W), 3 * W,
                                          ptr(o), W, B, R, n_heads, HEAD_PAD, scale, ptr(off), ptr(kw), ptr(ws),
                                          stream_ptr()), 'gvd_flash_attn_padded_f32')
    return o


class CompactIndex:
    """Device-side row maps of the masked-proposal compaction (gvd_compact_index; csrc/compact.hip)."""

    def __init__(self, pnt_mask):
        B, R1 = pnt_mask.shape
        R = R1 - 1
        dev = pnt_mask.device
        assert pnt_mask.dtype == torch.uint8 and pnt_mask.is_contiguous()
        self.B, self.R, self.cap = B, R, B * (R + 1)
        self.off = torch.empty(B + 1, dtype=torch.int32, device=dev)
        self.nvalid = torch.empty(B, dtype=torch.int32, device=dev)
        self.src_row = torch.zeros(self.cap, dtype=torch.int32, device=dev)   # (rows past the live count: row 0)
        self.cidx = torch.empty(B * R, dtype=torch.int32, device=dev)
        self.rep_w = torch.empty(B, dtype=torch.float32, device=dev)
        self.cmask = torch.zeros(self.cap, dtype=torch.uint8, device=dev)
        check(lib().gvd_compact_index(C.c_void_p(pnt_mask.data_ptr() + 1), R1, B, R, ptr(self.off), ptr(self.nvalid),
                                      ptr(self.src_row), ptr(self.cidx), ptr(self.rep_w), ptr(self.cmask), stream_ptr()),
              'gvd_compact_index')
        self.m_dev = self.off[B:]                                  # live compact rows, on the device

    def gather(self, x):
        """[B,R,D] dense -> [cap, D] compact (rows past the live count are not written)."""
        D = x.shape[-1]
        x2 = x.reshape(-1, D)
        out = torch.empty(self.cap, D, device=x.device, dtype=torch.float32)
        check(lib().gvd_gather_rows_f32(ptr(x2), x2.stride(0), ptr(self.src_row), ptr(out), D, D, self.cap, ptr(self.m_dev),
                                        stream_ptr()), 'gvd_gather_rows_f32')
        return out

    def expand(self, xc):
        """[cap, D] compact -> [B,R,D] dense (masked rows all receive their segment's representative row)."""
        D = xc.shape[-1]
        out = torch.empty(self.B, self.R, D, device=xc.device, dtype=torch.float32)
        check(lib().gvd_gather_rows_f32(ptr(xc), xc.stride(0), ptr(self.cidx), ptr(out), D, D, self.B * self.R, None,
                                        stream_ptr()), 'gvd_gather_rows_f32')
        return out


def fc_feature(segs_feat, num, w_seg, b_seg, pad_to=32, eps=1e-5):
    """model.py:306-308 in one launch (gvd_fc_feature): [layer_norm(mean_t segs_feat) | layer_norm(relu(seg_info_embed(num[:, 3:7])))
    | zero pad to a multiple of `pad_to` columns] -> [B, ldo].  segs_feat f32 [B,Ft,D] contiguous, num i64 [B,7]."""
    require_cuda_f32(segs_feat, w_seg, b_seg)
    B, Ft, D = segs_feat.shape
    S = w_seg.shape[0]
    assert segs_feat.is_contiguous() and num.dtype == torch.int64 and num.is_contiguous() and num.shape == (B, 7)
    assert w_seg.is_contiguous() and w_seg.shape[1] == 4
    ldo = -(-(D + S) // pad_to) * pad_to
    out = torch.empty(B, ldo, device=segs_feat.device, dtype=torch.float32)
    check(lib().gvd_fc_feature(ptr(segs_feat), ptr(num), ptr(w_seg), ptr(b_seg), ptr(out), B, Ft, D, S, ldo, eps, stream_ptr()),
          'gvd_fc_feature')
    return out


def loc_features(ppls, src_row, rows_dev, rows, n_frames, ldo=32):
    """model.py:357-360 on the compacted row set (gvd_loc_features): [x1,y1,x2,y2]/720, frame/T, zero pad -> [rows, ldo]."""
    require_cuda_f32(ppls)
    p2 = ppls.reshape(-1, ppls.shape[-1])
    assert p2.is_contiguous() and p2.shape[1] == 7
    out = torch.empty(rows, ldo, device=ppls.device, dtype=torch.float32)
    check(lib().gvd_loc_features(ptr(p2), ptr(src_row), ptr(rows_dev), ptr(out), rows, ldo, float(n_frames), stream_ptr()),
          'gvd_loc_features')
    return out


def affine_relu_rows_(x, scale, shift):
    """In place x = relu(x * scale + shift) over the last axis (BatchNorm1d in eval mode + ReLU; gvd_affine_relu_rows)."""
    require_cuda_f32(x, scale, shift)
    assert x.is_contiguous() and scale.is_contiguous() and shift.is_contiguous()
    D = x.shape[-1]
    check(lib().gvd_affine_relu_rows(ptr(x), ptr(scale), ptr(shift), x.numel() // D, D, stream_ptr()), 'gvd_affine_relu_rows')
    return x


def zero_rows_outside_window_(x, sample_idx):
    """In place x[b, t, :] = 0 for t outside [sample_idx[b,0], sample_idx[b,1]) (model.py:303-305,401)."""
    require_cuda_f32(x)
    B, Ft, D = x.shape
    assert x.is_contiguous() and sample_idx.dtype == torch.int64 and sample_idx.is_contiguous() and sample_idx.shape == (B, 2)
    check(lib().gvd_zero_rows_outside_window(ptr(x), ptr(sample_idx), B, Ft, D, stream_ptr()), 'gvd_zero_rows_outside_window')
    return x


def check_masked_rows_zero(x, pnt_mask, flag):
    """flag (int32 [1], device) |= 1 when a masked row of x [B,R,D] is not all-zero."""
    B, R, D = x.shape
    assert x.is_contiguous()
    check(lib().gvd_check_masked_rows_zero(ptr(x), D, C.c_void_p(pnt_mask.data_ptr() + 1), pnt_mask.shape[1], B, R,
                                           ptr(flag), stream_ptr()), 'gvd_check_masked_rows_zero')
