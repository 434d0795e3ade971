"""Device check of the EXPERIMENTAL direct-to-LDS variant of the pipelined GEMM (-DGVD_PIPE_LDSDMA=1, csrc/gemm_pipe.hip):
bitwise comparison with the product library on the preamble shapes + edge shapes (ragged M / N, device-side row count,
fused row gather, two K segments), then timing.  `build` needs no GPU.

    python tools/gemm_ldsdma_check.py build && gpurun -- python tools/gemm_ldsdma_check.py run
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, 'tools', '_bin')
SRC = os.path.join(ROOT, 'grounded-video-description_amd', 'csrc')
LIB = os.path.join(BIN, 'libgemm_regstaged.so')        # -DGVD_PIPE_LDSDMA=0: the register-staged form (reference for the plain products)
LIB_T = os.path.join(BIN, 'libgemm_ldsdma_t.so')        # -DGVD_PIPE_LDSDMA_T=1: EXPERIMENTAL direct loads for K-strided operands too


def build():
    os.makedirs(BIN, exist_ok=True)
    for out, flag in ((LIB, '-DGVD_PIPE_LDSDMA=0'), (LIB_T, '-DGVD_PIPE_LDSDMA_T=1')):
        cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-shared',
               '-Wno-unused-result', flag] + [os.path.join(SRC, f) for f in (
                   'gemm_f32.hip', 'gemm_pipe.hip', 'gemm_small.hip', 'gemv_f32.hip')] + ['-o', out]
        subprocess.run(cmd, check=True)
        print('built', out)


def run():
    import torch
    import gvd_amd  # noqa: F401
    from gvd_amd import hip
    GemmArgs, GemmSeg = hip.GemmArgs, hip.GemmSeg      # (the classes hip.lib() bound its argtypes to)
    # `old` = register-staged build, `new` = the product library (direct-to-LDS default)
    old = C.CDLL(LIB)
    old.gvd_gemm_nt_f32.restype = C.c_int
    old.gvd_gemm_nt_f32.argtypes = [C.POINTER(GemmArgs), C.c_void_p]
    new = hip.lib()
    exp = C.CDLL(LIB_T)
    exp.gvd_gemm_nt_f32.restype = C.c_int
    exp.gvd_gemm_nt_f32.argtypes = [C.POINTER(GemmArgs), C.c_void_p]
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device='cuda').manual_seed(1)
    rn = lambda *s: torch.randn(*s, device='cuda', generator=g)

    def args(A, W, b, out, act=0, m_dev=None, rmap=None, segs=None):
        a = GemmArgs()
        if segs is None:
            segs = [(A, W)]
        a.nseg = len(segs)
        for i, (x, w) in enumerate(segs):
            a.seg[i] = GemmSeg(x.data_ptr(), x.stride(0), 0, w.data_ptr(), w.stride(0), 0, x.shape[1])
        a.nbias = b.data_ptr() if b is not None else None
        a.C = out.data_ptr(); a.ldc = out.stride(0)
        a.M, a.N, a.batch, a.act = out.shape[0], out.shape[1], 1, act
        if m_dev is not None:
            a.m_dev = m_dev.data_ptr()
        if rmap is not None:
            a.a_row_map, a.a_src_rows = rmap.data_ptr(), A.shape[0]
        return a

    cases = []
    for (M, N, K) in ((40000, 2048, 2048), (33000, 1024, 2784), (33000, 3168, 1024), (33000, 512, 1024), (32999, 1020, 96),
                      (70001, 448, 2048)):
        A, W, b = rn(M, K), rn(N, K) / K ** 0.5, rn(N)
        cases.append(('M=%d N=%d K=%d' % (M, N, K), lambda o, A=A, W=W, b=b: args(A, W, b, o, 1), (M, N)))
    A, W, b = rn(50000, 1024), rn(1024, 1024) / 32, rn(1024)
    md = torch.tensor([41234], dtype=torch.int32, device='cuda')
    cases.append(('m_dev 41234 of 50000', lambda o: args(A, W, b, o, 0, m_dev=md), (50000, 1024)))
    src = rn(60000, 2048)
    rmap = torch.randperm(60000, device='cuda', generator=g)[:45000].sort()[0].to(torch.int32).contiguous()
    W2, b2 = rn(2048, 2048) / 45, rn(2048)
    cases.append(('row gather 45000 of 60000', lambda o: args(src, W2, b2, o, 1, m_dev=md, rmap=rmap), (45000, 2048)))
    X1, X2, V1, V2 = rn(33000, 512), rn(33000, 1024), rn(1024, 512) / 22, rn(1024, 1024) / 32
    cases.append(('two K segments 512 + 1024', lambda o: args(None, None, None, o, 0, segs=[(X1, V1), (X2, V2)]), (33000, 1024)))
    ok = True
    for name, mk, shape in cases:
        o1 = torch.zeros(*shape, device='cuda'); o2 = torch.zeros(*shape, device='cuda')
        a1, a2 = mk(o1), mk(o2)
        r1 = old.gvd_gemm_nt_f32(C.byref(a1), st()); r2 = new.gvd_gemm_nt_f32(C.byref(a2), st())
        torch.cuda.synchronize()
        same = bool(torch.equal(o1, o2))
        ok &= same and r1 == 0 and r2 == 0
        print('%-30s rc %d/%d  bitwise equal: %s%s' % (name, r1, r2, same, '' if same else '  max|diff| %.3e' % float((o1 - o2).abs().max())), flush=True)
    print('plain products, product library vs register-staged build:', 'ALL EQUAL' if ok else 'MISMATCH', flush=True)

    # ---- EXPERIMENTAL K-strided direct loads (-DGVD_PIPE_LDSDMA_T=1) vs the product library (register-staged K-strided path)
    def dx_args(dY, Wt, out):            # dX[M,K] = dY[M,N] @ W[N,K]
        a = GemmArgs()
        a.nseg = 1
        a.seg[0] = GemmSeg(dY.data_ptr(), dY.shape[1], 0, Wt.data_ptr(), Wt.shape[1], 0, dY.shape[1])
        a.C = out.data_ptr(); a.ldc = out.shape[1]
        a.M, a.N, a.batch, a.act = dY.shape[0], Wt.shape[1], 1, 0
        a.w_kstrided = 1
        return a

    def dw_args(dY, X, part, S):         # dW[N,K] = dY[M,N]^T @ X[M,K], split over S chunks of the contraction
        Mc = dY.shape[0] // S
        a = GemmArgs()
        a.nseg = 1
        a.seg[0] = GemmSeg(dY.data_ptr(), dY.shape[1], Mc * dY.shape[1], X.data_ptr(), X.shape[1], Mc * X.shape[1], Mc)
        a.C = part.data_ptr(); a.ldc = X.shape[1]; a.c_batch_stride = dY.shape[1] * X.shape[1]
        a.M, a.N, a.batch, a.act = dY.shape[1], X.shape[1], S, 0
        a.a_kstrided = a.w_kstrided = 1
        return a
    okt = True
    tcases = []
    dY, Wt = rn(33000, 1024), rn(1024, 2784) / 32
    tcases.append(('dX 33000 x 2784 (N=1024)', lambda o: dx_args(dY, Wt, o), (33000, 2784)))
    dY2, Wt2 = rn(32996, 3456), rn(3456, 1020) / 58
    tcases.append(('dX ragged 32996 x 1020 (N=3456)', lambda o: dx_args(dY2, Wt2, o), (32996, 1020)))
    dY3, X3 = rn(32768, 2048), rn(32768, 2048)
    tcases.append(('dW 2048 x 2048 (M=32768, S=4)', lambda o: dw_args(dY3, X3, o, 4), (4 * 2048, 2048)))
    dY4, X4 = rn(16384, 448), rn(16384, 2044)
    tcases.append(('dW ragged 448 x 2044 (M=16384, S=8)', lambda o: dw_args(dY4, X4, o, 8), (8 * 448, 2044)))
    for name, mk, shape in tcases:
        o1 = torch.zeros(*shape, device='cuda'); o2 = torch.zeros(*shape, device='cuda')
        a1, a2 = mk(o1), mk(o2)
        r1 = new.gvd_gemm_nt_f32(C.byref(a1), st()); r2 = exp.gvd_gemm_nt_f32(C.byref(a2), st())
        torch.cuda.synchronize()
        same = bool(torch.equal(o1, o2))
        okt &= same and r1 == 0 and r2 == 0
        print('%-40s rc %d/%d  bitwise equal: %s%s' % (name, r1, r2, same, '' if same else '  max|diff| %.3e' % float((o1 - o2).abs().max())), flush=True)
    print('K-strided products, experimental direct loads vs product library:', 'ALL EQUAL' if okt else 'MISMATCH', flush=True)
    for name, mk, shape, fl in (('dX 64000 x 2048 (N=2048)', None, None, None),):
        dYb, Wb = rn(64000, 2048), rn(2048, 2048) / 45
        ob = torch.empty(64000, 2048, device='cuda')
        a = dx_args(dYb, Wb, ob)
        res = []
        for lib_ in (new, exp):
            for _ in range(3):
                lib_.gvd_gemm_nt_f32(C.byref(a), st())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(15):
                lib_.gvd_gemm_nt_f32(C.byref(a), st())
            e1.record(); e1.synchronize()
            res.append(2.0 * 64000 * 2048 * 2048 * 15 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
        print('%s: register-staged %.1f TF/s | direct-to-LDS (experimental) %.1f TF/s' % (name, res[0], res[1]), flush=True)
    for (M, N, K) in ((256000, 2048, 2048), (205000, 1024, 2784), (205000, 3168, 1024), (205000, 1024, 1056), (205000, 1024, 512)):
        A, W, b = rn(M, K), rn(N, K) / K ** 0.5, rn(N)
        o = torch.empty(M, N, device='cuda')
        a = args(A, W, b, o, 1)
        res = []
        for lib_ in (old, new):
            for _ in range(3):
                lib_.gvd_gemm_nt_f32(C.byref(a), st())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(15):
                lib_.gvd_gemm_nt_f32(C.byref(a), st())
            e1.record(); e1.synchronize()
            res.append(2.0 * M * N * K * 15 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
        print('M=%d N=%d K=%d: register-staged %.1f TF/s | direct-to-LDS %.1f TF/s' % (M, N, K, res[0], res[1]), flush=True)
        del A, W, o


if __name__ == '__main__':
    (build if sys.argv[1:] == ['build'] else run)()
