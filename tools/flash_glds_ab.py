"""A/B of the K / V tile staging of the padded-head flash attention kernel (csrc/flash_attn_pad.hip): through registers
(-DGVD_FLASH_GLDS=0) vs direct global -> LDS loads (-DGVD_FLASH_GLDS=1).  Builds the one source file twice into
tools/_bin/ (hipcc is on the GPU box: same image), runs both on the same inputs - dense B x 1000 rows and the ragged
compacted-preamble shape - and reports bitwise equality + HIP-event times, interleaved so that clock drift hits both.
    python tools/flash_glds_ab.py [B]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

SRC = os.path.join(ROOT, 'grounded-video-description_amd', 'csrc', 'flash_attn_pad.hip')
BIN = os.path.join(ROOT, 'tools', '_bin')
os.makedirs(BIN, exist_ok=True)
libs = {}
for v in (0, 1):
    out = os.path.join(BIN, 'libflash_glds%d.so' % v)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-shared',
                           '-Wno-unused-result', '-DGVD_FLASH_GLDS=%d' % v, SRC, '-o', out])
    lib = C.CDLL(out)
    lib.gvd_flash_attn_padded_f32.restype = C.c_int
    lib.gvd_flash_attn_padded_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gvd_flash_attn_train_fwd_f32.restype = C.c_int
    lib.gvd_flash_attn_train_fwd_f32.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                 C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_float, C.c_uint64, C.c_void_p]
    libs[v] = lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nh, HP, R = 6, 176, 1000
dev = 'cuda'
W = nh * HP


def packed(rows):
    qkv = torch.zeros(rows, 3, nh, HP, device=dev)
    qkv[..., :171] = torch.randn(rows, 3, nh, 171, device=dev) * 0.5
    return qkv.reshape(rows, 3 * W)


def run(lib, qkv, o, Bn, Rn, off=None, kw=None):
    st = torch.cuda.current_stream().cuda_stream
    base = qkv.data_ptr()
    rc = lib.gvd_flash_attn_padded_f32(base, base + 4 * W, base + 8 * W, 3 * W, o.data_ptr(), W, Bn, Rn, nh, HP, 1.0 / 32,
                                       None if off is None else off.data_ptr(), None if kw is None else kw.data_ptr(), st)
    assert rc == 0, rc


def timed(f, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


# dense
qkv = packed(B * R)
outs = {v: torch.zeros(B * R, W, device=dev) for v in libs}
for v, lib in libs.items():
    run(lib, qkv, outs[v], B, R)
torch.cuda.synchronize()
print('dense B=%d R=%d: outputs bitwise equal: %s (max |diff| %.3e)' % (B, R, torch.equal(outs[0], outs[1]),
                                                                       float((outs[0] - outs[1]).abs().max())), flush=True)
fl = B * nh * 4.0 * R * R * HP
for rnd in range(3):
    for v, lib in libs.items():
        ms = timed(lambda: run(lib, qkv, outs[v], B, R))
        print('  round %d GLDS=%d: %.3f ms  %.1f TF/s incl. pads' % (rnd, v, ms, fl / ms / 1e9), flush=True)
# ragged
g = torch.Generator().manual_seed(0)
nv = (800 + torch.randint(-25, 26, (B,), generator=g)).tolist()
off = [0]
for n in nv:
    off.append(off[-1] + n + 1)
qc = packed(off[-1])
offd = torch.tensor(off, dtype=torch.int32, device=dev)
kw = torch.log2(torch.tensor([float(R - n) for n in nv], device=dev))
outs = {v: torch.zeros(off[-1], W, device=dev) for v in libs}
for v, lib in libs.items():
    run(lib, qc, outs[v], B, R + 1, offd, kw)
torch.cuda.synchronize()
print('ragged B=%d rows~801: outputs bitwise equal: %s' % (B, torch.equal(outs[0], outs[1])), flush=True)
fl = sum(nh * 4.0 * (n + 1) * (n + 1) * HP for n in nv)
for rnd in range(3):
    for v, lib in libs.items():
        ms = timed(lambda: run(lib, qc, outs[v], B, R + 1, offd, kw))
        print('  round %d GLDS=%d: %.3f ms  %.1f TF/s incl. pads' % (rnd, v, ms, fl / ms / 1e9), flush=True)
# training forward (B=64, Rp=1024, dropout): both stagings
Bt, Rp = 64, 1024
qt = packed(Bt * Rp)
res = {}
for v, lib in libs.items():
    o = torch.zeros(Bt * Rp, W, device=dev)
    lse = torch.zeros(Bt * nh, Rp, device=dev)
    f = lambda: lib.gvd_flash_attn_train_fwd_f32(qt.data_ptr(), 3 * W, o.data_ptr(), W, lse.data_ptr(), Bt, Rp, R, nh, HP, 1.0 / 32,
                                                 None, 0.2, 99, torch.cuda.current_stream().cuda_stream)
    assert f() == 0
    torch.cuda.synchronize()
    ms = timed(f)
    res[v] = (o, lse)
    print('train fwd B=%d GLDS=%d: %.3f ms' % (Bt, v, ms), flush=True)
print('train fwd outputs bitwise equal: %s, lse equal: %s' % (torch.equal(res[0][0], res[1][0]), torch.equal(res[0][1], res[1][1])))
