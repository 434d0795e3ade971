"""Build libgvd_hip.so (the C-ABI HIP library, include/gvd_hip.h) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the repo snapshot.
"""
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libgvd_hip.so')
STAMP = LIB + '.srchash'
SOURCES = ['gemm_f32.hip', 'gemv_f32.hip', 'attention.hip', 'vocab.hip', 'decode.hip', 'targets.hip', 'prof.hip', 'backward.hip', 'gru.hip', 'rowwise.hip', 'flash_attn.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-fno-gpu-rdc',
         '-Wno-unused-result']


def _source_hash():
    h = hashlib.sha256()
    files = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, 'gvd_common.h'), os.path.join(CSRC, 'gemv_f32.h'),
                                                       os.path.join(HERE, '..', 'include', 'gvd_hip.h')]
    for f in files:
        with open(f, 'rb') as fh:
            h.update(fh.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def library_path():
    return LIB


def is_fresh():
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return False
    with open(STAMP) as f:
        return f.read().strip() == _source_hash()


def build_library(force=False, verbose=True):
    if not force and is_fresh():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ['-o', LIB]
    if verbose:
        print('[gvd build]', ' '.join(cmd))
    subprocess.check_call(cmd)
    with open(STAMP, 'w') as f:
        f.write(_source_hash())
    return LIB


if __name__ == '__main__':
    build_library(force=True)
