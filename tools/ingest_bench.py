"""Ingest throughput on the GPU box: segments/s from page-cache-resident feature files to model-ready device tensors.
  (a) reference-style: per segment np.zeros + copy + masked_fill on the CPU (oracle/ingest_oracle.py), collate, .cuda()
  (b) gvd_amd.ingest.InferenceIngest: valid rows -> pinned staging (thread pool) -> async H2D -> zero fill on the GPU
python tools/ingest_bench.py [n_segments] [batch] [workers]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gvd_amd
from gvd_amd import ingest
from oracle import ingest_oracle as IO

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
BS = int(sys.argv[2]) if len(sys.argv) > 2 else 64
WS = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else '16').split(',')]
W = WS[0]
FT = int(os.environ.get('FT', '480'))
opt = gvd_amd.opts.default_opt(t_attn_size=FT)
root = tempfile.mkdtemp(dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
fr, sr, recs = IO.write_synthetic_dataset(root, opt, n_videos=(N + 3) // 4, segs_per_video=(4,), seed=1, num_frm=(300, 480, 600),
                                           short_props=False)
recs = recs[:N]
print('dataset: %d segments, %.1f MB region features each' % (len(recs), 1000 * 2048 * 4 / 1e6))
dev = torch.device('cuda', 0)
torch.set_num_threads(W)
# (a) reference-style CPU assembly, then one blocking H2D per tensor
t0 = time.perf_counter()
n = 0
for i in range(0, min(N, 2 * BS), BS):
    b = IO.assemble_batch(recs[i:i + BS], fr, sr, opt)
    d = {k: v.cuda() for k, v in b.items()}
    n += len(recs[i:i + BS])
torch.cuda.synchronize()
ta = time.perf_counter() - t0
print('(a) reference-style CPU padding/masking + H2D : %.1f segments/s' % (n / ta))
# (b) pipeline, for every worker count asked for
for W, numa in [(w, n) for n in (True, False) for w in WS]:
    ing = ingest.InferenceIngest(opt, fr, sr, device=dev, max_batch=BS, workers=W, numa_local=numa)
    for _ in ing.batches(recs[:BS], BS):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for rep in range(3):
        for chunk, t in ing.batches(recs, BS):
            n += len(chunk)
    torch.cuda.synchronize()
    tb = time.perf_counter() - t0
    print('(b) pinned staging (%d threads, %s) + async H2D + GPU zero fill, Ft=%d: %.1f segments/s (%.2f GB/s of features)'
          % (W, 'node-local' if numa else 'unpinned', FT, n / tb, n / tb * (1000 * 2048 * 4 + min(FT, 480) * 3072 * 4) / 1e9),
          flush=True)
    del ing
import shutil; shutil.rmtree(root, ignore_errors=True)
