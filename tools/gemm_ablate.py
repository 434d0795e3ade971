"""Where does the idle tenth of the matrix pipe go in the REGISTER-STAGED form of gemm_pipe_kernel (-DGVD_PIPE_LDSDMA=0; the
study that led to the direct-to-LDS default)?  Builds variants of the GEMM library with parts of
the pipelined kernel removed (-DGVD_PIPE_ABL=n: wrong results, unchanged MFMA work) and times the fc7-shaped product with
each.  `build` (no GPU needed) writes tools/_bin/libgemm_abl<n>.so; `run` (GPU) loads them through ctypes.

    python tools/gemm_ablate.py build && gpurun -- python tools/gemm_ablate.py run
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, 'tools', '_bin')
SRC = os.path.join(ROOT, 'grounded-video-description_amd', 'csrc')
MODES = {0: 'the kernel', 1: 'no k-tile barrier', 2: 'no LDS write pass', 3: 'no barrier, no write pass',
         4: 'no global loads after tile 0', 7: 'MFMA + fragment reads only', 8: 'no epilogue', 15: 'MFMA + fragment reads, no epilogue'}


def build():
    os.makedirs(BIN, exist_ok=True)
    for n in MODES:
        out = os.path.join(BIN, 'libgemm_abl%d.so' % n)
        cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-shared',
               '-Wno-unused-result', '-DGVD_PIPE_LDSDMA=0', '-DGVD_PIPE_ABL=%d' % n] + [os.path.join(SRC, f) for f in (
                   'gemm_f32.hip', 'gemm_pipe.hip', 'gemm_small.hip', 'gemv_f32.hip')] + ['-o', out]
        subprocess.run(cmd, check=True)
        print('built', out, flush=True)


def run():
    import torch
    import gvd_amd  # noqa: F401
    from gvd_amd.hip import GemmArgs, GemmSeg
    M, N, K = 256000, 2048, 2048
    A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
    out = torch.empty(M, N, device='cuda')
    fl = 2.0 * M * N * K
    for n, what in MODES.items():
        lib = C.CDLL(os.path.join(BIN, 'libgemm_abl%d.so' % n))
        lib.gvd_gemm_nt_f32.restype = C.c_int
        lib.gvd_gemm_nt_f32.argtypes = [C.POINTER(GemmArgs), C.c_void_p]
        g = GemmArgs()
        g.nseg = 1
        g.seg[0] = GemmSeg(A.data_ptr(), K, 0, W.data_ptr(), K, 0, K)
        g.nbias = b.data_ptr()
        g.C = out.data_ptr(); g.ldc = N
        g.M, g.N, g.batch, g.act = M, N, 1, 1
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(3):
            assert lib.gvd_gemm_nt_f32(C.byref(g), st) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.gvd_gemm_nt_f32(C.byref(g), st)
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print('abl=%2d %-36s %.3f ms  %.1f TF/s' % (n, what, ms, fl / ms / 1e9), flush=True)


if __name__ == '__main__':
    (build if sys.argv[1:] == ['build'] else run)()
