"""Run a tool script against a VARIANT build of libgvd_hip.so: the same sources compiled with extra hipcc flags (e.g. a
compile-time switch under test) into tools/_bin/<tag>/ - the product library and its stamp are not touched.
    python tools/with_cflags.py <tag> "<extra cflags>" <script.py> [args ...]
e.g. python tools/with_cflags.py variant1 "-DSOME_SWITCH=1" tools/profile_attn.py 256 10 5
GVD_VARIANT_CSRC=<dir>: compile the variant from another copy of csrc/ (A/B against an earlier commit without a switch in the sources).
(the sources keep no such switches once a measurement is decided: round 4 used it for the Newton step of the score tanh, the
ablations of the backward maps kernel and the two flash workgroup shapes - profiles/r04/*_ab_*.log, bwd_maps_ablate_b.log)"""
import importlib
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag, flags, script = sys.argv[1], sys.argv[2].split(), sys.argv[3]
b = importlib.import_module('grounded-video-description_amd.build')
d = os.path.join(ROOT, 'tools', '_bin', tag)
os.makedirs(d, exist_ok=True)
b.OBJ = os.path.join(d, 'obj')
b.LIB = os.path.join(d, 'libgvd_hip.so')
b.STAMP = b.LIB + '.srchash'
b.CFLAGS = list(b.CFLAGS) + flags
for f in os.environ.get('GVD_VARIANT_NOEXTRA', '').split(','):      # drop the per-file extra flags of build.EXTRA for these sources
    if f:
        b.EXTRA = {k: v for k, v in b.EXTRA.items() if k != f}
if os.environ.get('GVD_VARIANT_CSRC'):        # build the variant from ANOTHER copy of csrc/ (e.g. `git archive` of an earlier commit)
    b.CSRC = os.path.abspath(os.environ['GVD_VARIANT_CSRC'])
b.build_library(verbose=False)
print('[with_cflags] %s: %s' % (tag, ' '.join(flags)), flush=True)
sys.argv = [script] + sys.argv[4:]
runpy.run_path(script, run_name='__main__')
