"""Feature ingest: from the reference's on-disk feature layout to the tensors the model
receives — the six of `model(..., 'sample')` (InferenceIngest) and the eleven of `model(..., 'MLE')` (TrainIngest:
captions -> input_seq / gt_seq, GT boxes, box and frame masks; dataloader_anet.py:212-334) — as a per-rank
pinned-memory H2D pipeline.

Contract reproduced (dataloader_anet.py:175-212,317-354 + default collate + main.py:339-347):
  <feature_root>/<seg_id>.npy            f32 [T, P, 2048]   fc6 region features  -> ppls_feat [B, Rb, 2048]
  <seg_feature_root>/<vid[2:]>_resnet.npy f32 [F, 2048]  +  _bn.npy f32 [F, 1024] -> segs_feat [B, Ft, 3072]
  proposals [n, 7] (x1,y1,x2,y2,frame,cls,score; the reference reads them from an h5 file) -> ppls [B, Rb, 7]
  pnt_mask u8 [B, Rb+1] = [0 | score <= prop_thresh (| cls == 0) | 1 beyond n], masked rows of ppls / ppls_feat zeroed,
  Rb = max(1, max_b n_b) (main.py:339-341), frames beyond F zero, sample_idx = clip(round(F * t / dur), 0, Ft),
  num i64 [B,7] = LongTensor copy of [1, n, 0, seg_idx, n_seg, t0/dur, t1/dur].

Split of work (MI355X-first): host threads only copy the VALID raw rows of the (memory-mapped) files into pinned
staging and make the byte masks; one async H2D per tensor slice on a copy stream; the zero padding / masked-row
zeroing — three full passes over 8 MB per sample on the CPU in the reference — happens on the GPU in place
(`gvd_zero_masked_rows`, touches only the rows it clears).  Staging is double-buffered so batch i+1 is read and copied
while batch i is being decoded.  All compute is in the HIP library: there is no CPU fallback for the device half.
"""
import ctypes as C
import os
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import hip, ops


def _read_rows_into(path, dst, max_rows):
    """Read the first min(rows, max_rows) rows of a [..., D] float32 .npy straight into `dst`, a float32 [>= rows, D] view
    of the pinned staging buffer whose rows may be a COLUMN BLOCK of wider rows (the frame features `<vid>_resnet.npy
    [F,2048]` / `_bn.npy [F,1024]` are the two column blocks of segs_feat's 3072-wide rows, dataloader_anet.py:198-206).
    ONE native call per file (gvd_npy_read_rows_f32: open, header parse, pread - scatter preadv for strided rows - from the
    page cache into the pinned rows): no mmap page faults, no intermediate array, no Python header parsing or file object,
    GIL released.  Returns (rows_read, rows_in_file)."""
    assert dst.strides[1] == 4 and dst.dtype == np.float32
    n = C.c_int64(0)
    rc = hip.lib().gvd_npy_read_rows_f32(os.fsencode(path), dst.ctypes.data, min(max_rows, dst.shape[0]), dst.shape[1],
                                         dst.strides[0], C.byref(n))
    if rc < 0:
        if rc <= -1000:
            raise ValueError('%s: not a C-ordered float32 .npy with last dimension %d (reader code %d)' % (path, dst.shape[1], rc))
        raise OSError(-rc, os.strerror(-rc), path)
    return n.value, rc


def _local_node_cpus():
    """CPUs of the NUMA node the calling thread runs on (intersected with the process's affinity mask), or None when the
    topology cannot be read.  The pinned staging buffers and - for page-cache-resident files written by this process - the
    file pages are first-touched from the constructing thread, so readers on the same node copy node-locally; readers
    spread over both sockets of the 256-thread GPU host were measured SLOWER the more of them there were."""
    try:
        cpu = os.sched_getcpu()
        allowed = os.sched_getaffinity(0)
        base = '/sys/devices/system/node'
        for d in os.listdir(base):
            if not d.startswith('node') or not d[4:].isdigit():
                continue
            cpus = set()
            with open(os.path.join(base, d, 'cpulist')) as f:
                for part in f.read().strip().split(','):
                    if not part:
                        continue
                    a, _, b = part.partition('-')
                    cpus.update(range(int(a), int(b or a) + 1))
            if cpu in cpus:
                cpus &= allowed
                return cpus or None
    except (OSError, ValueError, AttributeError):
        pass
    return None


class _Slot:
    """One set of pinned staging buffers (sized for the largest batch)."""

    def __init__(self, max_batch, R, Ft, att_feat, fc_feat, pin):
        def buf(*shape, dtype=torch.float32):
            t = torch.empty(*shape, dtype=dtype)
            return t.pin_memory() if pin else t
        self.feat = buf(max_batch, R, att_feat)
        self.segs = buf(max_batch, Ft, fc_feat)
        self.ppls = buf(max_batch, R, 7)
        self.mask = buf(max_batch, R + 1, dtype=torch.uint8)
        self.fmask = buf(max_batch, Ft, dtype=torch.uint8)
        self.num = buf(max_batch, 7, dtype=torch.int64)
        self.sidx = buf(max_batch, 2, dtype=torch.int64)
        # numpy views of the same memory for the host half (no per-segment tensor -> array conversions)
        self.feat_np, self.segs_np, self.ppls_np = self.feat.numpy(), self.segs.numpy(), self.ppls.numpy()
        self.mask_np, self.fmask_np = self.mask.numpy(), self.fmask.numpy()
        self.num_np, self.sidx_np = self.num.numpy(), self.sidx.numpy()
        self.n_pps = [0] * max_batch
        self.n_frm = [0] * max_batch
        self.B = 0
        self.free = None            # event: the previous upload from this slot has finished reading it


def cpu_budget():
    """CPUs this process may actually burn: the smaller of its affinity mask and its container's CFS quota (cgroup v2
    cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us).  The GPU box shows 256 CPUs and grants 16: 32 busy reader threads
    next to the launching main thread ran the container into its quota, the kernel throttled every thread for the rest of the
    100 ms period, and the files -> captions timeline showed it as all readers stalling ~50 ms at once every few batches
    (profiles/r05/files_read_path.txt)."""
    try:
        n = float(len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        n = float(os.cpu_count() or 8)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, float(quota) / float(period))
    except (OSError, ValueError):
        try:
            quota = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            period = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if quota > 0 and period > 0:
                n = min(n, quota / period)
        except (OSError, ValueError):
            pass
    return max(1.0, n)


def default_workers():
    """Reader threads of one staging call.  One segment is three page-cache -> pinned-memory copies (8 + 4 + 2 MB at Ft =
    480); 8 to 16 mapped-copy readers keep the staging of a batch of 64 under the GPU's time for it (28 vs 31.7 ms, profiles/r05)
    and more only help while the CPU budget holds: four CPUs stay free for the launching thread, the staging thread and the
    HIP runtime's own threads."""
    return int(max(2, min(16, cpu_budget() - 4)))


class InferenceIngest:
    """records: dicts with seg_id '<vid>_segment_<k>', n_seg_in_vid, timestamps (t0, t1), duration, proposals [n,7]."""

    def __init__(self, opt, feature_root, seg_feature_root, device=None, max_batch=256, exclude_bgd_det=False,
                 workers=None, depth=2, numa_local=None):
        if numa_local is None:
            numa_local = os.environ.get('GVD_INGEST_NUMA', '1') == '1'
        if workers is None:
            workers = default_workers()
        self.opt = opt
        self.feature_root, self.seg_feature_root = feature_root, seg_feature_root
        self.device = device
        self.exclude_bgd_det = exclude_bgd_det
        self.R = opt.num_sampled_frm * opt.num_prop_per_frm          # max_proposal, dataloader_anet.py:45
        self.Ft = opt.t_attn_size
        pin = device is not None
        self.slots = [_Slot(max_batch, self.R, self.Ft, opt.att_feat_size, opt.fc_feat_size, pin) for _ in range(depth)]
        self.max_batch = max_batch
        cpus = _local_node_cpus() if numa_local else None

        def pin_worker(cpus=cpus):
            if cpus:
                try:
                    os.sched_setaffinity(0, cpus)         # (0 = the calling THREAD on Linux)
                except OSError:
                    pass
        self.workers = workers
        # 'mapped' (default): map each file, copy its rows in user space; 'pread': read() into the staging rows - for feature
        # files another process may truncate while they are read (csrc/ingest.hip, GVD_READ_*)
        how = os.environ.get('GVD_INGEST_READ', 'mapped')
        if how not in ('mapped', 'pread'):
            raise ValueError("GVD_INGEST_READ must be 'mapped' or 'pread', got %r" % how)
        self.read_mode = 1 if how == 'mapped' else 0
        # ONE staging thread (pinned to the staging buffers' NUMA node; the native reader threads it spawns per batch inherit
        # its affinity).  It holds the GIL only for the per-record bookkeeping below - the file reads of a whole batch are one
        # GIL-free native call - so it does not fight the main thread, which is busy enqueueing the previous batch's launches.
        self._outer = ThreadPoolExecutor(max_workers=1, initializer=pin_worker)
        self.copy_stream = torch.cuda.Stream(device=device) if device is not None else None
        self._next = 0
        self._tls = threading.local()
        self.trace = None           # a list: every batch appends its host time stamps + copy-stream events (tools/files_timeline.py)

    # ------------------------------------------------------------------ host half
    def _prep_one(self, slot, b, rec):
        """Everything of one record that does not need a file: proposals, byte mask, `num`.  -> (n proposals, n_pps)"""
        opt = self.opt
        props = np.asarray(rec['proposals'], dtype=np.float64)
        n = props.shape[0]
        n_pps = min(n, self.R)
        masked = props[:, 6] <= opt.prop_thresh                                                   # l.194-196
        if self.exclude_bgd_det:
            masked |= props[:, 5] == 0
        slot.ppls_np[b, :n_pps] = props[:n_pps]                           # (float64 -> float32 rounding, as .float())
        m = slot.mask_np[b]
        m[0] = 0                                                          # legacy pad column, main.py:345
        m[1:1 + n_pps] = masked[:n_pps]
        m[1 + n_pps:] = 1
        t0, t1 = rec['timestamps']
        dur = rec['duration']
        seg_idx = rec['seg_id'].split('_segment_')[1]
        # main.py copies the FloatTensor `num` into a LongTensor: float32 rounding, then the two time stamps truncate
        slot.num_np[b] = np.array([1, n_pps, 0, int(seg_idx), rec['n_seg_in_vid'], t0 * 1. / dur, t1 * 1. / dur],
                                  dtype=np.float32).astype(np.int64)
        return n, n_pps

    def stage(self, records):
        """Fill the next staging slot from the feature files: per-record bookkeeping on this thread, then ALL file reads of
        the batch (3 per segment: region features, resnet / bn frame features - the two column blocks of segs_feat's
        3072-wide rows, scatter-read straight into the pinned rows) as ONE native call on `workers` native threads."""
        assert 0 < len(records) <= self.max_batch
        tr = {'stage_begin': time.perf_counter()} if self.trace is not None else None
        slot = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        if slot.free is not None:
            slot.free.synchronize()                 # its previous upload still reads the pinned buffers
            slot.free = None
        if tr is not None:
            tr['slot_free'] = time.perf_counter()
        B = slot.B = len(records)
        nj = 3 * B
        paths = (C.c_char_p * nj)()
        dsts = (C.c_void_p * nj)()
        max_rows, Dd, stride = (C.c_int64 * nj)(), (C.c_int64 * nj)(), (C.c_int64 * nj)()
        rows_read, rows_file = (C.c_int64 * nj)(), (C.c_int64 * nj)()
        counts = []
        feat_base, feat_step = slot.feat_np.ctypes.data, slot.feat_np.strides[0]
        segs_base, segs_step, segs_row = slot.segs_np.ctypes.data, slot.segs_np.strides[0], slot.segs_np.strides[1]
        for b, rec in enumerate(records):
            counts.append(self._prep_one(slot, b, rec))
            seg_id = rec['seg_id']
            vid = seg_id.split('_segment_')[0]
            for k, (pth, dst, mr, d, st) in enumerate((
                    (os.path.join(self.feature_root, seg_id + '.npy'), feat_base + b * feat_step, self.R,
                     slot.feat_np.shape[2], slot.feat_np.strides[1]),
                    (os.path.join(self.seg_feature_root, vid[2:] + '_resnet.npy'), segs_base + b * segs_step, self.Ft, 2048, segs_row),
                    (os.path.join(self.seg_feature_root, vid[2:] + '_bn.npy'), segs_base + b * segs_step + 2048 * 4, self.Ft,
                     slot.segs_np.shape[2] - 2048, segs_row))):
                j = 3 * b + k
                paths[j], dsts[j], max_rows[j], Dd[j], stride[j] = os.fsencode(pth), dst, mr, d, st
        if tr is not None:
            tr['prep_done'] = time.perf_counter()
        job_ns = (C.c_int64 * nj)() if tr is not None else None
        failed = hip.lib().gvd_npy_read_batch_f32(paths, dsts, max_rows, Dd, stride, nj, self.workers, self.read_mode,
                                                  rows_read, rows_file, job_ns)
        if tr is not None:
            tr['read_done'] = time.perf_counter()
            tr['job_ms_sum'], tr['job_ms_max'] = sum(job_ns) / 1e6, max(job_ns) / 1e6
        if failed:
            for j in range(nj):
                rc = rows_file[j]
                if rc < 0:
                    pth = os.fsdecode(paths[j])
                    if rc <= -1000:
                        raise ValueError('%s: not a C-ordered float32 .npy with last dimension %d (reader code %d)' % (pth, Dd[j], rc))
                    raise OSError(-rc, os.strerror(-rc), pth)
        for b, rec in enumerate(records):
            n, n_pps = counts[b]
            assert n == rows_file[3 * b], 'proposal count does not match the region feature file'          # l.191
            assert n_pps == rows_read[3 * b]
            n_frm, num_frm = rows_read[3 * b + 1], rows_file[3 * b + 1]
            fm = slot.fmask_np[b]
            fm[:n_frm] = 0
            fm[n_frm:] = 1
            t0, t1 = rec['timestamps']
            dur = rec['duration']
            sidx = np.array([np.round(num_frm * t0 * 1. / dur), np.round(num_frm * t1 * 1. / dur)])   # l.207-208
            slot.sidx_np[b] = np.clip(np.round(sidx), 0, self.Ft).astype(np.int64)
            slot.n_pps[b], slot.n_frm[b] = n_pps, n_frm
        if tr is not None:
            tr['stage_end'] = time.perf_counter()
            slot.trace = tr
        return slot

    # ------------------------------------------------------------------ device half
    def upload(self, slot):
        """Async H2D of the valid rows + in-place zero fill on the copy stream.  Returns the dict of device tensors;
        the calling stream is made to wait for the upload."""
        if self.device is None:
            raise RuntimeError('InferenceIngest.upload needs a GPU (the padding/masking half runs in the HIP library)')
        B, dev = slot.B, self.device
        Rb = max(max(slot.n_pps[:B]), 1)                                   # main.py:339-341
        cur = torch.cuda.current_stream(dev)
        feat = torch.empty(B, Rb, self.opt.att_feat_size, device=dev)
        segs = torch.empty(B, self.Ft, self.opt.fc_feat_size, device=dev)
        ppls = torch.empty(B, Rb, 7, device=dev)
        mask = torch.empty(B, Rb + 1, dtype=torch.uint8, device=dev)
        fmask = torch.empty(B, self.Ft, dtype=torch.uint8, device=dev)
        num = torch.empty(B, 7, dtype=torch.int64, device=dev)
        sidx = torch.empty(B, 2, dtype=torch.int64, device=dev)
        cs = self.copy_stream
        cs.wait_stream(cur)                        # the allocator may hand out blocks earlier kernels still use
        tr = getattr(slot, 'trace', None) if self.trace is not None else None
        if tr is not None:
            tr['upload_enqueue'] = time.perf_counter()
            tr['h2d_start'] = torch.cuda.Event(enable_timing=True)
            tr['h2d_end'] = torch.cuda.Event(enable_timing=True)
            tr['h2d_start'].record(cs)
        with torch.cuda.stream(cs):
            mask.copy_(slot.mask[:B, :Rb + 1], non_blocking=True)
            fmask.copy_(slot.fmask[:B], non_blocking=True)
            num.copy_(slot.num[:B], non_blocking=True)
            sidx.copy_(slot.sidx[:B], non_blocking=True)
            # valid rows only, as few copies as the layout allows: a run of segments whose rows are all valid is ONE
            # contiguous chunk of the pinned buffer and of the device tensor (the ANet-Entities files hold T x 100 proposals
            # for every segment, so normally the whole batch is one copy per tensor); ragged segments go one by one
            def runs(counts, full):
                b = 0
                while b < B:
                    e = b
                    while e < B and counts[e] == full:
                        e += 1
                    if e > b:
                        yield b, e, full
                        b = e
                    else:
                        yield b, b + 1, counts[b]
                        b += 1
            for b0, b1, n in runs(slot.n_pps, self.R if Rb == self.R else -1):
                if n:
                    if b1 - b0 > 1 or n == self.R == Rb:
                        feat[b0:b1].copy_(slot.feat[b0:b1], non_blocking=True)
                        ppls[b0:b1].copy_(slot.ppls[b0:b1], non_blocking=True)
                    else:
                        feat[b0, :n].copy_(slot.feat[b0, :n], non_blocking=True)
                        ppls[b0, :n].copy_(slot.ppls[b0, :n], non_blocking=True)
            for b0, b1, f in runs(slot.n_frm, self.Ft):
                if f:
                    if f == self.Ft:
                        segs[b0:b1].copy_(slot.segs[b0:b1], non_blocking=True)
                    else:
                        segs[b0, :f].copy_(slot.segs[b0, :f], non_blocking=True)
            extra = self._device_extras(slot, ppls, Rb, dev)      # needs the proposals BEFORE masked rows are zeroed
            ops.zero_masked_rows(feat, mask, mask_off=1)
            ops.zero_masked_rows(ppls, mask, mask_off=1)
            ops.zero_masked_rows(segs, fmask)
            slot.free = torch.cuda.Event()
            slot.free.record(cs)
            if tr is not None:
                tr['h2d_end'].record(cs)
                tr['upload_enqueued'] = time.perf_counter()
                self.trace.append(tr)
        cur.wait_stream(cs)
        out = dict(segs_feat=segs, num=num, ppls=ppls, ppls_feat=feat, sample_idx=sidx, pnt_mask=mask)
        out.update(extra)
        return out

    def _device_extras(self, slot, ppls, Rb, dev):
        return {}

    def batches(self, records, batch_size):
        """Yield (records_of_batch, tensors) with the files of batch i+1 being read while batch i is consumed."""
        chunks = [records[i:i + batch_size] for i in range(0, len(records), batch_size)]
        if not chunks:
            return
        pending = self.pool_stage(chunks[0])
        for i, ch in enumerate(chunks):
            slot = pending.result()
            tensors = self.upload(slot)
            if i + 1 < len(chunks):
                pending = self.pool_stage(chunks[i + 1])
            yield ch, tensors

    def pool_stage(self, records):
        # the one staging thread keeps the order of the slots; the reads of a batch fan out inside its native call
        return self._outer.submit(self.stage, records)


class TrainIngest(InferenceIngest):
    """The training half of the loader contract (dataloader_anet.py:212-334 + main.py:213-232) on top of the feature
    pipeline: records additionally carry `caption` = the reference's caption-file entry {'caption': [words], 'clss':
    [[class names] per box], 'idx': [[word positions] per box], 'bbox': [[x1,y1,x2,y2]], 'frm_idx': [frame per box]};
    `vocab` = dict(wtoi: word -> index (str or int), wtod: class name -> 1-based detection index).
    Emits, besides the six inference tensors: seq i64 [B,1,L+1,4] (col 0 = word id, or vocab_size + class for grounded
    words; col 1 = 1/2 whether the class name is the word; col 2 = class; col 3 = word id of grounded words),
    gt_seq i64 [B,10,L], gt_boxes f32 [B,NB,6], mask_boxes u8 [B,1,NB,L+1] (0 at a box's word position),
    frm_mask u8 [B,Rb,NB] (1 = proposal and box on different frames) and num[:,2] = number of boxes; NB = batch
    maximum (main.py:216-218).  The caption/box tensors are tiny and are built on the host; the [Rb,NB] frame mask is
    formed on the GPU from the uploaded proposal frames — before the masked proposal rows are zeroed, as in the
    reference (l.333 precedes l.343)."""

    MAX_GT_BOX = 100                                                   # dataloader_anet.py:44

    def __init__(self, opt, feature_root, seg_feature_root, vocab, **kw):
        super().__init__(opt, feature_root, seg_feature_root, **kw)
        self.wtoi = {w: int(i) for w, i in vocab['wtoi'].items()}
        self.wtod = dict(vocab['wtod'])

    def _caption_arrays(self, rec):
        L, V = self.opt.seq_length, self.opt.vocab_size
        cap = rec['caption']
        words = cap['caption']
        # one annotation per (box, label) whose word position is inside the caption window, numbered in file order
        ann = [(pos, serial, b, name) for serial, (b, name, pos) in enumerate(
            (b, name, cap['idx'][b][j]) for b, names in enumerate(cap['clss']) for j, name in enumerate(names)
            if cap['idx'][b][j] < L)]
        ann.sort(key=lambda a: a[0])                                   # by word position, stable
        keep = []
        for pos, serial, b, name in ann:
            x1, y1, x2, y2 = cap['bbox'][b]
            if not self.opt.test_mode and not ((x2 - x1 + 1) != 1 and (y2 - y1 + 1) != 1):
                continue                                               # zero-area boxes are dropped (l.244-248)
            keep.append((pos, serial, b, name))
        seq = np.zeros((1, L + 1, 4), dtype=np.int64)
        gts = np.zeros((10, L), dtype=np.int64)
        n_w = min(len(words), L)
        ids = np.array([self.wtoi[w] for w in words[:n_w]], dtype=np.int64)
        seq[0, 1:1 + n_w, 0] = ids
        gts[0, :n_w] = ids
        for pos, serial, b, name in keep:                              # later annotations of a position overwrite earlier ones
            d = self.wtod[name]
            seq[0, pos + 1] = (V + d, (name != words[pos]) + 1, d, ids[pos])
        n_box = min(len(keep), self.MAX_GT_BOX)
        boxes = np.zeros((n_box, 6), dtype=np.float32)
        bmask = np.ones((1, n_box, L + 1), dtype=np.uint8)
        for i, (pos, serial, b, name) in enumerate(keep[:n_box]):
            if self.opt.test_mode:
                boxes[i] = (0, 0, 0, 0, -1, self.wtod[name])
            else:
                boxes[i] = tuple(cap['bbox'][b]) + (cap['frm_idx'][b], self.wtod[name])
            bmask[0, i, pos + 1] = 0
        return seq, gts, boxes, bmask

    def stage(self, records):
        slot = super().stage(records)
        slot.train = [self._caption_arrays(r) for r in records]
        for b, t in enumerate(slot.train):
            slot.num[b, 2] = t[2].shape[0]
        return slot

    def _device_extras(self, slot, ppls, Rb, dev):
        B, L = slot.B, self.opt.seq_length
        NB = max(max(t[2].shape[0] for t in slot.train), 1)            # main.py:216-218
        seq = torch.from_numpy(np.stack([t[0] for t in slot.train]))
        gts = torch.from_numpy(np.stack([t[1] for t in slot.train]))
        boxes = torch.zeros(B, NB, 6)
        bmask = torch.ones(B, 1, NB, L + 1, dtype=torch.uint8)
        nb = torch.zeros(B, dtype=torch.int64)
        for b, t in enumerate(slot.train):
            k = t[2].shape[0]
            boxes[b, :k] = torch.from_numpy(t[2])
            bmask[b, :, :k] = torch.from_numpy(t[3])
            nb[b] = k
        d = lambda x: x.to(dev, non_blocking=True)
        seq, gts, boxes, bmask, nb = d(seq), d(gts), d(boxes), d(bmask), d(nb)
        npps = d(torch.tensor(slot.n_pps[:B], dtype=torch.int64))
        r_ok = torch.arange(Rb, device=dev).view(1, Rb, 1) < npps.view(B, 1, 1)
        k_ok = torch.arange(NB, device=dev).view(1, 1, NB) < nb.view(B, 1, 1)
        differ = ppls[:, :, 4].unsqueeze(2) != boxes[:, :, 4].unsqueeze(1)
        frm = torch.where(r_ok & k_ok, differ, torch.ones_like(differ)).to(torch.uint8)
        return dict(seq=seq, gt_seq=gts, gt_boxes=boxes, mask_boxes=bmask, frm_mask=frm)
