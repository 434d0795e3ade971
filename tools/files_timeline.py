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
    mode = os.environ.get('GVD_TL_MODE', '')
    if mode == 'noupload':
        # (diagnosis E3) the staging thread reads every batch's files as usual, but the model is fed ONE resident batch: no H2D
        resident = next(iter(ing.batches(recs[:B], B)))[1]
        res = tuple(resident[k] for k in keys)
        real_upload = ing.upload

        def fake_upload(slot):
            slot.free = None
            if ing.trace is not None and getattr(slot, 'trace', None) is not None:
                tr = slot.trace
                tr['upload_enqueue'] = tr['upload_enqueued'] = time.perf_counter()
                for k in ('h2d_start', 'h2d_end'):
                    tr[k] = torch.cuda.Event(enable_timing=True)
                    tr[k].record()
                ing.trace.append(tr)
            return dict(zip(keys, res))
        ing.upload = fake_upload
    if mode == 'memcpy':
        # (diagnosis E1) the reads come from user-space mappings of the same files (np.load(mmap_mode='r'), prefaulted): plain
        # memcpy from the page cache, no read() system call
        import numpy as np
        from concurrent.futures import ThreadPoolExecutor
        maps = {}
        for r_ in recs:
            sid = r_['seg_id']
            vid = sid.split('_segment_')[0]
            for pth in (os.path.join(fr, sid + '.npy'), os.path.join(sr, vid[2:] + '_resnet.npy'), os.path.join(sr, vid[2:] + '_bn.npy')):
                if pth not in maps:
                    maps[pth] = np.load(pth, mmap_mode='r')
                    float(np.asarray(maps[pth]).reshape(-1)[::1024].sum())     # prefault
        pool = ThreadPoolExecutor(32)
        real_stage = ing.stage

        def mem_stage(records):
            t_b = time.perf_counter()
            slot = ing.slots[ing._next]
            ing._next = (ing._next + 1) % len(ing.slots)
            if slot.free is not None:
                slot.free.synchronize()
                slot.free = None
            t_f = time.perf_counter()
            slot.B = len(records)

            def one(br):
                b, rec = br
                ing._prep_one(slot, b, rec)
                sid = rec['seg_id']
                vid = sid.split('_segment_')[0]
                f = maps[os.path.join(fr, sid + '.npy')].reshape(-1, 2048)
                np.copyto(slot.feat_np[b, :f.shape[0]], f)
                a_, b_ = maps[os.path.join(sr, vid[2:] + '_resnet.npy')], maps[os.path.join(sr, vid[2:] + '_bn.npy')]
                n = min(a_.shape[0], ing.Ft)
                np.copyto(slot.segs_np[b, :n, :2048], a_[:n])
                np.copyto(slot.segs_np[b, :n, 2048:], b_[:n])
                slot.fmask_np[b, :n] = 0
                slot.fmask_np[b, n:] = 1
                slot.sidx_np[b] = (0, n)
                slot.n_pps[b], slot.n_frm[b] = f.shape[0], n
            list(pool.map(one, enumerate(records)))
            t_e = time.perf_counter()
            if ing.trace is not None:
                slot.trace = {'stage_begin': t_b, 'slot_free': t_f, 'prep_done': t_f, 'read_done': t_e, 'stage_end': t_e,
                              'job_ms_sum': 0.0, 'job_ms_max': 0.0}
            return slot
        ing.stage = mem_stage
    ing.trace, trace = [], []
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.cuda.synchronize()
    sampler = None
    if os.environ.get('GVD_TL_SAMPLE'):
        # (diagnosis) where are this process's threads while the pipeline runs?  state / wchan / kernel stack of every task, every
        # ~5 ms, from a helper PROCESS (no GIL, no perturbation of the threads it looks at)
        import subprocess
        sampler = subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'proc_stack_sampler.py'),
                                    str(os.getpid()), '1.2', os.environ['GVD_TL_SAMPLE']])
        time.sleep(0.3)                # (interpreter start-up of the sampler)
    def cpu_times():
        out = {}
        for t in os.listdir('/proc/self/task'):
            try:
                st = open('/proc/self/task/%s/stat' % t).read()
                f = st[st.rindex(')') + 2:].split()
                out[int(t)] = (int(f[11]), int(f[12]), int(f[36]))          # utime, stime (clock ticks), last CPU
            except OSError:
                pass
        return out
    def cgroup_cpu():
        # CPU quota of this container and how often / how long the kernel throttled it (cgroup v2, then v1 layout)
        out = {}
        for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu.stat', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us',
                  '/sys/fs/cgroup/cpu/cpu.cfs_period_us', '/sys/fs/cgroup/cpu/cpu.stat', '/sys/fs/cgroup/cpuset.cpus.effective',
                  '/sys/fs/cgroup/cpuset/cpuset.cpus'):
            try:
                out[f] = ' '.join(open(f).read().split())
            except OSError:
                pass
        return out

    def sched():
        out = {}
        for t in os.listdir('/proc/self/task'):
            try:
                a, b, c = open('/proc/self/task/%s/schedstat' % t).read().split()
                out[int(t)] = (int(a), int(b), int(c))               # ns on a CPU, ns runnable but waiting for one, time slices
            except (OSError, ValueError):
                pass
        return out
    print('CPUs: os.cpu_count %s, affinity of this thread %d' % (os.cpu_count(), len(os.sched_getaffinity(0))))
    g0 = cgroup_cpu()
    s0 = sched()
    c0 = cpu_times()
    t0 = time.perf_counter()
    model.sample_pipelined(produce(recs), {'max_in_flight': mif, 'trace': trace})
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    c1 = cpu_times()
    s1 = sched()
    g1 = cgroup_cpu()
    for f in g1:
        print('cgroup %s: before [%s] after [%s]' % (f, g0.get(f), g1[f]))
    ds = sorted(((s1[t][0] - s0.get(t, (0, 0, 0))[0], s1[t][1] - s0.get(t, (0, 0, 0))[1], s1[t][2] - s0.get(t, (0, 0, 0))[2], t)
                 for t in s1), reverse=True)
    print('schedstat of the threads over the run (ms on a CPU / ms runnable-but-waiting / slices, tid):')
    print('  ' + '  '.join('%.0f/%.0f/%d(t%d)' % (a / 1e6, b / 1e6, c, t) for a, b, c, t in ds[:40]))
    print('  all threads: on a CPU %.0f ms, waiting for a CPU %.0f ms' % (sum(x[0] for x in ds) / 1e6, sum(x[1] for x in ds) / 1e6))
    tick = os.sysconf('SC_CLK_TCK')
    d = sorted(((c1[t][0] - c0.get(t, (0, 0, 0))[0], c1[t][1] - c0.get(t, (0, 0, 0))[1], t, c1[t][2]) for t in c1), reverse=True)
    print('CPU time of this process\'s threads over the %.0f ms run (user ms, kernel ms, tid, last cpu; main tid %d); threads born '
          'during the run count from 0:' % (1e3 * (t1 - t0), os.getpid()))
    print('  ' + '  '.join('%d/%d(t%d c%d)' % (1e3 * u / tick, 1e3 * k / tick, t, c) for u, k, t, c in d[:14]))
    print('  all threads: user %d ms, kernel %d ms; threads alive at the end %d' % (1e3 * sum(x[0] for x in d) / tick,
                                                                                 1e3 * sum(x[1] for x in d) / tick, len(c1)))
    if sampler is not None:
        sampler.wait()
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
