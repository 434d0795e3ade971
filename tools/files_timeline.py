"""Host + device timeline of the files -> captions flow (synthetic split on disk -> ingest.InferenceIngest ->
TopDownModel.sample_pipelined): per batch, when the staging thread read its files, when the main thread obtained / enqueued it,
and when its upload, preamble and token loop ran on the GPU (stream events).  All times in ms from the start of the run.
    python tools/files_timeline.py [n_segments=512] [batch=64] [max_in_flight=3] [depth=2]"""
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gvd_amd  # noqa: E402,F401
from gvd_amd import att_model, ingest, opts, synth  # noqa: E402

n_seg = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
mif = int(sys.argv[3]) if len(sys.argv) > 3 else 3
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dev = torch.device('cuda', 0)
root = tempfile.mkdtemp(prefix='gvd_tl_', dir='/dev/shm')
try:
    opt = opts.default_opt(vocab_size=5000, t_attn_size=480)
    fr, sr, recs = synth.write_feature_split(root, opt, n_seg, seed=3)
    model = att_model.TopDownModel(opt)
    model.load_state_dict(synth.init_state_dict(opt, seed=15, profile='trained_like'))
    model = model.to(dev).eval()
    ing = ingest.InferenceIngest(opt, fr, sr, device=dev, max_batch=B, depth=depth,
                                 workers=int(os.environ['GVD_TL_WORKERS']) if os.environ.get('GVD_TL_WORKERS') else None)
    keys = ('segs_feat', 'ppls', 'num', 'ppls_feat', 'sample_idx', 'pnt_mask')

    def produce(rs):
        for chunk, t in ing.batches(rs, B):
            yield tuple(t[k] for k in keys)
    model.sample_pipelined(produce(recs[:2 * B]), {'max_in_flight': mif})
    torch.cuda.synchronize()
    ing.trace, trace = [], []
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.sample_pipelined(produce(recs), {'max_in_flight': mif, 'trace': trace})
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    st = torch.cuda.memory_stats()
    print('allocator: device allocs %d, frees %d, reserved %.1f GB, alloc retries %d'
          % (st['num_device_alloc'], st['num_device_free'], st['reserved_bytes.all.current'] / 1e9, st['num_alloc_retries']))
    print('%d segments, batch %d, max_in_flight %d, staging depth %d: %.1f ms = %.1f captions/s'
          % (n_seg, B, mif, depth, 1e3 * (t1 - t0), n_seg / (t1 - t0)))
    ms = lambda t: 1e3 * (t - t0)
    ev = lambda e: e0.elapsed_time(e)
    print('reader threads: %d' % ing.workers)
    print('batch | staging thread: begin slot_free prep_done read_done end [sum / max of the per-file read times] | main: got upload_enq enqueued throttled | GPU: h2d[start end] pre[start end] dec[start end]')
    for i, (a, b) in enumerate(zip(ing.trace, trace)):
        print('%3d | %7.1f %7.1f %7.1f %7.1f %7.1f [%7.1f %5.1f] | %7.1f %7.1f %7.1f %7.1f | %7.1f %7.1f  %7.1f %7.1f  %7.1f %7.1f'
              % (i, ms(a['stage_begin']), ms(a['slot_free']), ms(a['prep_done']), ms(a['read_done']), ms(a['stage_end']),
                 a['job_ms_sum'], a['job_ms_max'], ms(b['got']),
                 ms(a['upload_enqueued']), ms(b['enqueued']), ms(b['throttled']), ev(a['h2d_start']), ev(a['h2d_end']),
                 ev(b['pre_start']), ev(b['pre_end']), ev(b['dec_start']), ev(b['dec_end'])))
    if os.environ.get('GVD_TL_EXPERIMENT'):
        # (diagnosis) the staging thread ALONE - file reads into the pinned slots, no upload - while the main thread keeps the GPU
        # busy with ONE kind of work: which kind slows the host-side reads down?
        ing.trace = None
        x = torch.randn(8192, 8192, device=dev)
        big = torch.empty(1 << 28, device=dev)
        t = next(iter(ing.batches(recs[:B], B)))[1]
        d = tuple(t[k] for k in keys)
        hostbuf = torch.empty(64, 1000, 2048).pin_memory()
        devbuf = torch.empty(64, 1000, 2048, device=dev)
        P = {k: v.detach() for k, v in model._decode_params().items()}
        with torch.no_grad():
            pre0 = model._preamble(d[0], d[2], d[1], d[3], d[4], d[5], allow_compact=True)
        from gvd_amd import ops

        def w_none():
            time.sleep(0.02)

        def w_matmul():
            for _ in range(4):
                x @ x

        def w_devcopy():
            for _ in range(20):
                big.copy_(big.roll(1) if False else big)          # (device -> device, 1 GB each)
                big.add_(1.0)

        def w_h2d():
            for _ in range(4):
                devbuf.copy_(hostbuf, non_blocking=True)

        def w_preamble():
            with torch.no_grad():
                model._preamble(d[0], d[2], d[1], d[3], d[4], d[5], allow_compact=True)

        def w_decode():
            with torch.no_grad():
                ops.greedy_decode(pre0, P, pre0['pnt_mask'], model.seq_length, model.unk_idx, flags=model._flags())
        import threading
        for label, work in (('nothing', w_none), ('matmuls', w_matmul), ('device add_', w_devcopy), ('H2D copies', w_h2d),
                            ('preamble', w_preamble), ('token loop (greedy_decode)', w_decode), ('nothing again', w_none)):
            stop = [False]
            ts = []

            def stager():
                i = 0
                while not stop[0] and len(ts) < 12:
                    t0s = time.perf_counter()
                    ing._outer.submit(ing.stage, recs[(i % 8) * B:(i % 8 + 1) * B]).result()
                    ts.append(1e3 * (time.perf_counter() - t0s))
                    for sl in ing.slots:
                        sl.free = None
                    i += 1
            th = threading.Thread(target=stager)
            tb = time.perf_counter()
            th.start()
            n = 0
            while th.is_alive():
                work()
                n += 1
                if n % 4 == 0:
                    torch.cuda.synchronize()
            th.join()
            torch.cuda.synchronize()
            print('staging while the GPU runs %-28s: ms per batch %s' % (label, ' '.join('%.1f' % v for v in ts[2:])))
finally:
    shutil.rmtree(root, ignore_errors=True)
