"""Host half of the feature ingest (grounded-video-description_amd/ingest.py) against the restated reference dataloader
(oracle/ingest_oracle.py): the raw rows it stages, the byte masks, `num`, `sample_idx`.  The device half (zero fill of
padded / masked rows) is covered by tests/test_gpu_ingest.py; without a GPU `upload` must refuse to run."""
import numpy as np
import pytest
import torch

import os

import gvd_amd
from gvd_amd import ingest
from oracle import cases, ingest_oracle as IO, ref_harness


@pytest.fixture(scope='module')
def dataset(tmp_path_factory):
    opt = gvd_amd.opts.default_opt(t_attn_size=480)
    root = str(tmp_path_factory.mktemp('feats'))
    fr, sr, recs = IO.write_synthetic_dataset(root, opt, seed=3)
    return opt, fr, sr, recs


@pytest.mark.parametrize('exclude_bgd', [False, True])
def test_staging_matches_reference_dataloader(dataset, exclude_bgd):
    opt, fr, sr, recs = dataset
    want = IO.assemble_batch(recs, fr, sr, opt, exclude_bgd_det=exclude_bgd)
    ing = ingest.InferenceIngest(opt, fr, sr, device=None, max_batch=len(recs), exclude_bgd_det=exclude_bgd, workers=3)
    slot = ing.stage(recs)
    B = len(recs)
    assert slot.B == B
    assert torch.equal(slot.num[:B], want['num'])
    assert torch.equal(slot.sidx[:B], want['sample_idx'])
    assert torch.equal(slot.mask[:B], want['pnt_mask'])           # this batch holds a full-length sample: Rb == R
    for b in range(B):
        n, f = slot.n_pps[b], slot.n_frm[b]
        keep = want['pnt_mask'][b, 1:1 + n] == 0
        # valid, unmasked rows are exactly the reference's; masked / padded rows are the device kernel's job
        assert torch.equal(slot.feat[b, :n][keep], want['ppls_feat'][b, :n][keep])
        assert torch.equal(slot.ppls[b, :n][keep], want['ppls'][b, :n][keep])
        assert torch.equal(slot.segs[b, :f], want['segs_feat'][b, :f])
        assert int(slot.fmask[b, :f].sum()) == 0 and int(slot.fmask[b, f:].sum()) == opt.t_attn_size - f
        assert float(want['segs_feat'][b, f:].abs().sum()) == 0.0


def test_upload_needs_the_gpu(dataset):
    opt, fr, sr, recs = dataset
    ing = ingest.InferenceIngest(opt, fr, sr, device=None, max_batch=2)
    slot = ing.stage(recs[:2])
    with pytest.raises(RuntimeError):
        ing.upload(slot)


def test_sample_idx_rounding_is_half_to_even():
    """dataloader_anet.py:207 uses np.round (banker's rounding): F*t/dur = 2.5 -> 2, 3.5 -> 4."""
    assert np.round(2.5) == 2 and np.round(3.5) == 4


@pytest.mark.skipif(not ref_harness.reference_available(), reason='no /root/reference here')
@pytest.mark.parametrize('exclude_bgd', [False, True])
def test_oracle_equals_the_real_reference_dataloader(dataset, exclude_bgd):
    """THE PIN of oracle/ingest_oracle.py: the reference's real DataLoader.__getitem__ (dataloader_anet.py:175-354, run
    through oracle/ref_dataloader_harness.py) + main.py's trimming vs the restatement, all eleven tensors, bit for bit."""
    from oracle import ref_dataloader_harness as RH
    opt, fr, sr, recs = dataset
    vocab = IO.synthetic_vocab(opt)
    ds = RH.build_reference_dataset(recs, vocab, fr, sr, opt, exclude_bgd_det=exclude_bgd)
    for idx in (list(range(len(recs))), [2], [1, 4]):
        ref = RH.reference_batch(ds, idx, train=True)
        ours = IO.assemble_train_batch([recs[i] for i in idx], vocab, fr, sr, opt, exclude_bgd_det=exclude_bgd)
        for k in cases.INGEST_KEYS:
            assert ref[k].dtype == ours[k].dtype and tuple(ref[k].shape) == tuple(ours[k].shape), k
            assert torch.equal(ref[k], ours[k]), k
        inf = IO.assemble_batch([recs[i] for i in idx], fr, sr, opt, exclude_bgd_det=exclude_bgd)
        refi = RH.reference_batch(ds, idx, train=False)
        for k in ('segs_feat', 'num', 'ppls', 'ppls_feat', 'sample_idx', 'pnt_mask'):
            if k == 'num':      # eval batches carry the box count too (the loader does not know the mode); the model ignores it
                assert torch.equal(refi[k][:, [0, 1, 3, 4, 5, 6]], inf[k][:, [0, 1, 3, 4, 5, 6]])
            else:
                assert torch.equal(refi[k], inf[k]), k


@pytest.mark.parametrize('name', [n for n, s in cases.CASES.items() if s['mode'] == 'ingest'])
def test_oracle_matches_reference_dataloader_golden(name, golden_dir, tmp_path):
    """The travelling pin: tests/golden/ingest_*.npz holds the reference dataloader's outputs on the seeded dataset."""
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    opt, vocab, fr, sr, recs = cases.build_ingest_case(name, str(tmp_path))
    assert len(recs) == int(g['n_records'])
    ours = IO.assemble_train_batch(recs, vocab, fr, sr, opt)
    for k in cases.INGEST_KEYS:
        assert tuple(ours[k].shape) == tuple(g['shape_' + k]), k
        assert cases._bits_checksum(ours[k]) == int(g['fp_' + k]), k
        if k in cases.INGEST_SMALL:
            assert np.array_equal(ours[k].numpy(), g[k]), k


def test_train_ingest_host_half_matches_oracle(dataset):
    """TrainIngest's caption / box staging (product code) vs the pinned oracle."""
    opt, fr, sr, recs = dataset
    vocab = IO.synthetic_vocab(opt)
    want = IO.assemble_train_batch(recs, vocab, fr, sr, opt)
    ing = ingest.TrainIngest(opt, fr, sr, vocab, device=None, max_batch=len(recs), workers=2)
    slot = ing.stage(recs)
    assert torch.equal(slot.num[:len(recs)], want['num'])
    NB = want['gt_boxes'].shape[1]
    for b, (seq, gts, boxes, bmask) in enumerate(slot.train):
        k = boxes.shape[0]
        assert np.array_equal(seq, want['seq'][b].numpy()) and np.array_equal(gts, want['gt_seq'][b].numpy())
        assert np.array_equal(boxes, want['gt_boxes'][b, :k].numpy()) and float(want['gt_boxes'][b, k:].abs().sum()) == 0
        assert np.array_equal(bmask, want['mask_boxes'][b, :, :k].numpy()) and bool((want['mask_boxes'][b, :, k:] == 1).all())
        assert k <= NB


@pytest.mark.parametrize('mode', [0, 1])                       # GVD_READ_PREAD, GVD_READ_MAPPED (include/gvd_hip.h)
def test_native_batch_reader_modes(tmp_path, mode):
    """gvd_npy_read_batch_f32 (csrc/ingest.hip; replaces the np.load calls of dataloader_anet.py:189,198-199): both ways of
    moving a file's rows into strided destination rows give np.load's bytes - row caps, ragged files, more threads than jobs,
    a second call on the persistent reader threads - and report bad files per job instead of failing the batch: a file cut
    short of what its header promises (-1007, also for the mapped read, which checks the size BEFORE it maps), a float64
    file (-1003), a wrong last dimension (-1006), a missing file (-ENOENT)."""
    import ctypes
    import errno
    from gvd_amd import hip
    rng = np.random.default_rng(5)
    D = 96
    good = []
    for i, rows in enumerate((1, 7, 100, 2500)):
        a = rng.standard_normal((rows, D)).astype(np.float32)
        if i == 2:
            a = a.reshape(4, 25, D)                                # rows = product of all but the last dimension
        np.save(tmp_path / ('g%d.npy' % i), a)
        good.append((str(tmp_path / ('g%d.npy' % i)), a.reshape(-1, D)))
    whole = np.load(good[3][0])
    np.save(tmp_path / 'short.npy', whole)
    with open(tmp_path / 'short.npy', 'r+b') as f:
        f.truncate(128 + 2000 * D * 4)
    np.save(tmp_path / 'f64.npy', whole[:3].astype(np.float64))
    np.save(tmp_path / 'dim.npy', whole[:3, :D - 1].copy())
    bad = [(str(tmp_path / 'short.npy'), -1007), (str(tmp_path / 'f64.npy'), -1003), (str(tmp_path / 'dim.npy'), -1006),
           (str(tmp_path / 'missing.npy'), -errno.ENOENT)]
    jobs = [(p, a, cap) for (p, a), cap in zip(good, (4, 3, 100, 2048))] + [(p, None, 2048) for p, _ in bad]
    n = len(jobs)
    stride = (D + 8) * 4                                           # destination rows wider than the file's
    lib = hip.lib()
    for n_threads in (1, 3, 32):
        dst = [np.full((cap, D + 8), -7.0, np.float32) for _, _, cap in jobs]
        paths = (ctypes.c_char_p * n)(*[p.encode() for p, _, _ in jobs])
        dsts = (ctypes.c_void_p * n)(*[d.ctypes.data for d in dst])
        caps = (ctypes.c_int64 * n)(*[cap for _, _, cap in jobs])
        Ds = (ctypes.c_int64 * n)(*([D] * n))
        strides = (ctypes.c_int64 * n)(*([stride] * n))
        rows_read = (ctypes.c_int64 * n)()
        rows_file = (ctypes.c_int64 * n)()
        ns = (ctypes.c_int64 * n)()
        failed = lib.gvd_npy_read_batch_f32(paths, dsts, caps, Ds, strides, n, n_threads, mode, rows_read, rows_file, ns)
        assert failed == len(bad)
        for i, (p, a, cap) in enumerate(jobs):
            if a is None:
                assert rows_file[i] == bad[i - len(good)][1] and rows_read[i] == 0, (p, rows_file[i])
                continue
            k = min(cap, a.shape[0])
            assert rows_file[i] == a.shape[0] and rows_read[i] == k and ns[i] > 0
            assert np.array_equal(dst[i][:k, :D], a[:k])
            assert np.all(dst[i][:k, D:] == -7.0) and np.all(dst[i][k:] == -7.0)      # nothing outside the rows it was given
    assert lib.gvd_npy_read_batch_f32(paths, dsts, caps, Ds, strides, n, 2, 7, rows_read, rows_file, None) == -errno.EINVAL
    # the empty batch returns at once (it used to wait for -1 pooled threads forever, holding the pool's call lock) and
    # leaves the pool usable
    for n_threads in (0, 1, 4):
        assert lib.gvd_npy_read_batch_f32(paths, dsts, caps, Ds, strides, 0, n_threads, mode, rows_read, rows_file, None) == 0
    assert lib.gvd_npy_read_batch_f32(paths, dsts, caps, Ds, strides, n, 2, mode, rows_read, rows_file, None) == len(bad)


def test_read_mode_selection(dataset, monkeypatch):
    opt, fr, sr, recs = dataset
    assert ingest.InferenceIngest(opt, fr, sr, device=None, max_batch=2).read_mode == 1
    monkeypatch.setenv('GVD_INGEST_READ', 'pread')
    ing = ingest.InferenceIngest(opt, fr, sr, device=None, max_batch=2)
    assert ing.read_mode == 0 and ing.stage(recs[:2]).B == 2
    monkeypatch.setenv('GVD_INGEST_READ', 'bounce')
    with pytest.raises(ValueError):
        ingest.InferenceIngest(opt, fr, sr, device=None, max_batch=2)


def test_default_reader_threads_follow_the_cpu_budget(monkeypatch):
    """ingest.default_workers: sized to what the container may burn (affinity mask, CFS quota), not to os.cpu_count() - the GPU
    box shows 256 CPUs and grants 16."""
    import builtins
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == '/sys/fs/cgroup/cpu.max':
            import io
            return io.StringIO('1600000 100000\n')
        return real_open(path, *a, **k)
    monkeypatch.setattr(os, 'sched_getaffinity', lambda pid: set(range(256)))
    monkeypatch.setattr(builtins, 'open', fake_open)
    assert ingest.cpu_budget() == 16.0 and ingest.default_workers() == 12
    monkeypatch.setattr(os, 'sched_getaffinity', lambda pid: set(range(4)))
    assert ingest.cpu_budget() == 4.0 and ingest.default_workers() == 2


def test_reader_pool_survives_a_fork(tmp_path):
    """The native reader threads are created once per process; a fork()ed child inherits the pool object but none of its
    threads.  The child's first batch must build its own pool instead of waiting for helpers that do not exist there."""
    import ctypes
    from gvd_amd import hip
    a = np.arange(64 * 8, dtype=np.float32).reshape(64, 8)
    n = 6
    for i in range(n):
        np.save(tmp_path / ('f%d.npy' % i), a + i)

    def read_batch():
        lib = hip.lib()
        dst = [np.zeros((64, 8), np.float32) for _ in range(n)]
        paths = (ctypes.c_char_p * n)(*[str(tmp_path / ('f%d.npy' % i)).encode() for i in range(n)])
        dsts = (ctypes.c_void_p * n)(*[d.ctypes.data for d in dst])
        caps = (ctypes.c_int64 * n)(*([64] * n))
        Ds = (ctypes.c_int64 * n)(*([8] * n))
        strides = (ctypes.c_int64 * n)(*([32] * n))
        rr, rf = (ctypes.c_int64 * n)(), (ctypes.c_int64 * n)()
        failed = lib.gvd_npy_read_batch_f32(paths, dsts, caps, Ds, strides, n, 4, 1, rr, rf, None)
        return failed == 0 and all(np.array_equal(d, a + i) for i, d in enumerate(dst))

    assert read_batch()                       # the parent's pool now has threads
    pid = os.fork()
    if pid == 0:
        ok = False
        try:
            import signal
            signal.alarm(30)                  # a child that waits for threads it does not have must not hang the suite
            ok = read_batch()
        finally:
            os._exit(0 if ok else 1)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, status
    assert read_batch()                       # and the parent's pool is untouched
