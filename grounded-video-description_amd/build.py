"""Build libgvd_hip.so (the C-ABI HIP library, include/gvd_hip.h) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the repo snapshot.  Every source is compiled to its own
object (in parallel, only when it or a header changed), then linked.
"""
import fcntl
import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build')
LIB = os.path.join(HERE, 'libgvd_hip.so')
STAMP = LIB + '.srchash'
SOURCES = ['gemm_f32.hip', 'gemm_pipe.hip', 'gemm_small.hip', 'gemm_ks.hip', 'gemm_n192.hip', 'gemv_f32.hip', 'attention.hip', 'vocab.hip', 'decode.hip', 'decode_persistent.hip',
           'targets.hip', 'prof.hip', 'backward.hip', 'gru.hip', 'rowwise.hip', 'flash_attn_pad.hip', 'compact.hip', 'enc_attn_bwd.hip', 'ingest.hip', 'beam.hip', 'train_fused.hip', 'optim.hip', 'stream_mm.hip', 'gemm_dxs.hip', 'train_rows.hip', 'lstm_seq.hip']
HEADERS = ['gvd_common.h', 'gemm_common.h', 'gemv_f32.h', 'top2.h', 'decode_persistent.h', 'philox.h', 'enc_dropout.h']
CFLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wno-unused-result']
# decode_persistent.hip keeps ~150 weight registers per lane for the whole launch; the SLP vectorizer would pair its
# accumulators into v_pk_fma_f32 and splat every resident weight into a register PAIR (2x the footprint -> spills)
EXTRA = {'decode_persistent.hip': ['-fno-slp-vectorize']}
LDFLAGS = ['--offload-arch=gfx950', '-shared', '-fPIC', '-fno-gpu-rdc', '-Wl,-z,defs']     # (-z defs: an undefined symbol is a LINK error, not a dlopen failure on the GPU box)


def _header_hash():
    h = hashlib.sha256()
    for f in [os.path.join(CSRC, x) for x in HEADERS] + [os.path.join(HERE, '..', 'include', 'gvd_hip.h')]:
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def _file_hash(src, hh):
    h = hashlib.sha256()
    with open(os.path.join(CSRC, src), 'rb') as fh:
        h.update(fh.read())
    h.update(hh.encode())
    h.update(' '.join(CFLAGS + EXTRA.get(src, [])).encode())
    return h.hexdigest()


def _source_hash():
    hh = _header_hash()
    h = hashlib.sha256()
    for s in SOURCES:
        h.update(_file_hash(s, hh).encode())
    h.update(' '.join(LDFLAGS).encode())
    return h.hexdigest()


def library_path():
    return LIB


def is_fresh():
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return False
    with open(STAMP) as f:
        return f.read().strip() == _source_hash()


def build_library(force=False, verbose=True):
    """Compile + link in-tree.  Safe under `torch.distributed.run` (every rank may call this at import time): the whole
    build runs under an exclusive file lock, a rank that waited re-checks freshness instead of rebuilding, and the
    library / stamp are published with atomic renames so no process can dlopen a half-written file."""
    if not force and is_fresh():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and is_fresh():         # another process built it while we waited for the lock
                return LIB
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    hh = _header_hash()

    def compile_one(src):
        obj = os.path.join(OBJ, src + '.o')
        tag = obj + '.hash'
        want = _file_hash(src, hh)
        if not force and os.path.exists(obj) and os.path.exists(tag):
            with open(tag) as f:
                if f.read().strip() == want:
                    return obj
        cmd = [hipcc] + CFLAGS + EXTRA.get(src, []) + ['-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print('[gvd build]', ' '.join(cmd))
        subprocess.check_call(cmd)
        with open(tag, 'w') as f:
            f.write(want)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp = '%s.tmp.%d' % (LIB, os.getpid())
    cmd = [hipcc] + LDFLAGS + objs + ['-o', tmp]
    if verbose:
        print('[gvd build]', ' '.join(cmd))
    subprocess.check_call(cmd)
    if os.path.exists(STAMP):
        os.unlink(STAMP)                 # never a fresh stamp next to an old library
    os.replace(tmp, LIB)
    with open(STAMP + '.tmp', 'w') as f:
        f.write(_source_hash())
    os.replace(STAMP + '.tmp', STAMP)
    return LIB


if __name__ == '__main__':
    build_library(force=True)
