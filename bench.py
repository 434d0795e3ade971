"""bench.py — captions/sec of the GVD greedy-decode hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--t-attn Ft] [--vocab V]
                    [--mode sample|train] [--beam K --frames T]

One "step" = one full `'sample'` call (AttModel._sample: per-segment preamble + 20-token greedy loop)
over one batch of synthetic segments that is already resident in HBM.  With `--gpus N > 1` and no RANK in
the environment the script re-executes itself under `torch.distributed.run` (one process per GPU, rendezvous on
127.0.0.1); launched by the driver under torch.distributed.run it reads RANK/LOCAL_RANK/WORLD_SIZE.  The path
shards purely over the batch of video segments, so every rank decodes its own B segments with no data-path
collective (weak scaling) and the only communication is the barrier + max-reduction of the elapsed time
(`--mode train` adds the RCCL gradient all-reduce).

Prints ONE JSON line (rank 0): metric/value (whole-job captions/s), `roofline` of the dominant hand-written
kernel (the attention streaming kernel, timed live with HIP events on its stream over the timed region),
`cpu_baseline` (the CPU oracle = port of the reference path, timed on this box's host cores; the timing of the REAL
reference in the build container is attached from profiles/cpu_reference_timing.json) and — for the default
workload, whose inputs and weights are exactly the committed reference case `greedy_b256_v5000_ft10_trained` —
`parity`: the decoded ids / attended regions of the timed run compared with the reference's own output.
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL / tensor sharing fail with the legacy mode); the driver's
# environment exports it already - keep it for any environment this script is launched from
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
# every product / softmax of the path that would leave libgvd_hip.so for a torch library op raises instead of running
# (ops.library_fallback); the line reports the count of such calls as "library_gemms" (0, or the run would have failed)
os.environ.setdefault('GVD_STRICT', '1')

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s measured copy
MFMA_F32_PEAK_TFS = 157.3   # dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32; MI355X_MICROARCH.md) - no xf32 on gfx950
GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')
GOLDEN_B256 = os.path.join(GOLDEN_DIR, 'greedy_b256_v5000_ft10_trained.npz')


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--mode', choices=['sample', 'train'], default='sample',
                    help="sample: greedy decode captions/s (headline); train: 'MLE' fwd+bwd+Adam step segments/s")
    ap.add_argument('--batch', type=int, default=None, help='segments per GPU per step (default 256 sample / 64 train)')
    ap.add_argument('--t-attn', type=int, default=10, help='temporal positions Ft (BASELINE: [B,10,3072]; reference default 480)')
    ap.add_argument('--vocab', type=int, default=5000)
    ap.add_argument('--beam', type=int, default=1, help='beam size (1 = greedy; 5 = BASELINE configs[4])')
    ap.add_argument('--frames', type=int, default=10, help='sampled frames T (regions R = 100*T; configs[4] uses 20)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                    help="torch.distributed backend: 'nccl' (= RCCL over xGMI, the product path) or 'gloo' - the only way to run "
                         'N > 1 ranks on a box with ONE GPU (ranks share cuda:0; RCCL refuses two ranks on one device): used to '
                         'exercise the N-rank code path end to end where no multi-GPU node is available, never for numbers')
    ap.add_argument('--files', type=int, default=0, metavar='N',
                    help='only the integrated files -> captions measurement over a synthetic on-disk split of N segments at the '
                         'reference-default 480 temporal positions (ingest -> pipelined decode -> JSON writers); the default run '
                         'carries a 256-segment version as config.files_to_captions_ft480')
    ap.add_argument('--no-sections', action='store_true',
                    help='default line only: skip the short configs[2] (train B=64), configs[4] (beam=5 x 20 frames, B=64) and '
                         'Ft=480 (reference-default frame count, B=256) sections the default run appends to its JSON line')
    ap.add_argument('--no-dp-section', action='store_true',
                    help='skip the collective-safe configs[3] section (32 segments/GPU optimisation steps + all-reduce timing) '
                         'every sample-mode line carries')
    ap.add_argument('--overlap', action='store_true',
                    help='pipeline the K steps on two HIP streams (preamble of step i+1 || token loop of step i). Off by '
                         'default: co-scheduling stretches the attention kernel, so its live roofline figure would not '
                         'describe the kernel (measured gains: +2.5%% at B=256, +14%% at B=32, +19%% at B=4)')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--h2d', action='store_true',
                    help='PCIe-inclusive variant: every step first copies its batch from pinned host memory (double-buffered '
                         'on a copy stream, overlapped with the previous step); reported separately, never the headline value')
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: become `torch.distributed.run ... bench.py <same args>`."""
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.exit('bench.py: --gpus %d requested but this node exposes %d GPU(s) (torch.cuda.device_count()); '
                 'run with --gpus <= %d' % (args.gpus, have, max(have, 1)))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def _reference_timing():
    """Timing of the REAL reference (imported from /root/reference, which does not exist on the GPU box) measured in the
    build container by tools/time_reference_cpu.py and committed; attached to cpu_baseline for context."""
    p = os.path.join(ROOT, 'profiles', 'cpu_reference_timing.json')
    if os.path.exists(p):
        with open(p) as f:
            return json.load(f)
    return None


def _best_threads(fn):
    """torch's default of one thread per hardware thread is pathologically slow on many-core hosts for these small
    ops: try a few thread counts (one call each) and keep the fastest."""
    ncpu = os.cpu_count() or 1
    best = None
    for nt in sorted({min(ncpu, n) for n in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        fn()                                  # warm-up at this thread count
        t0 = time.time()
        fn()
        dt = time.time() - t0
        if best is None or dt < best[1]:
            best = (nt, dt)
        if dt > 20.0:
            break
    torch.set_num_threads(best[0])
    return best[0], ncpu


def _timed_loop(fn, seconds, max_n=200):
    n, t0 = 0, time.time()
    while True:
        fn()
        n += 1
        if time.time() - t0 >= seconds or n >= max_n:
            break
    return n, time.time() - t0


def cpu_baseline(opt, sd, seconds, beam=1, threads=None):
    """The oracle (CPU port of the reference path) on BASELINE configs[0]: B=4 greedy (or beam), same shapes.
    threads: skip the thread-count search (the short baselines of the config sections reuse the headline's choice)."""
    from gvd_amd import synth
    from oracle import gvd_oracle as O
    inp = synth.make_inputs(opt, 4, seed=0, train=False)
    a = [inp[k] for k in ('segs_feat', 'num', 'ppls', 'ppls_feat', 'sample_idx', 'pnt_mask')]
    with torch.no_grad():
        if beam > 1:
            run = lambda: O.sample_beam(sd, opt, *a, beam_size=beam)
        else:
            run = lambda: O.sample_greedy(sd, opt, *a)
        if threads:
            nt, ncpu = min(threads, os.cpu_count() or 1), os.cpu_count() or 1
            torch.set_num_threads(nt)
            run()                                 # warm-up
        else:
            nt, ncpu = _best_threads(run)
        n, dt = _timed_loop(run, seconds)
    return {'value': round(4 * n / dt, 3), 'unit': 'captions/s', 'cores': nt, 'kind': 'port', 'host_cpus': ncpu,
            'sample': '%d %s sample() calls of B=4 (L=20, %dx100 regions, Ft=%d, V=%d) with oracle/gvd_oracle.py '
                      '(torch-CPU restatement pinned bit-for-bit to the reference), %.1f s, %s'
                      % (n, 'beam=%d' % beam if beam > 1 else 'greedy', opt.num_sampled_frm, opt.t_attn_size,
                         opt.vocab_size, dt, '%d threads' % nt if threads else 'best of 8/16/32/64 threads'),
            'reference_in_build_container': None if threads else _reference_timing()}


def cpu_baseline_train(opt, sd, seconds, full=True):
    """Oracle 'MLE' forward + autograd backward + clip + Adam on B=8 segments (eval-mode arithmetic: dropout is the only
    difference to a train-mode step and costs nothing on CPU)."""
    from gvd_amd import synth
    from oracle import gvd_oracle as O
    Bc = 8
    inp = synth.trim_to_batch(synth.make_inputs(opt, Bc, seed=0, train=True))
    W = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v) for k, v in sd.items()}
    params = [v for v in W.values() if v.requires_grad]
    optim = torch.optim.Adam(params, lr=opt.learning_rate)

    def run():
        optim.zero_grad(set_to_none=True)
        lm, a2, gl, cl, _ = O.forward_train(W, opt, *[inp[k] for k in synth.FORWARD_ORDER])
        (lm + opt.w_att2 * a2 + opt.w_grd * gl + opt.w_cls * cl).backward()
        torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], opt.grad_clip)
        optim.step()
    ncpu = os.cpu_count() or 1
    nt = min(ncpu, 32)
    torch.set_num_threads(nt)
    run()
    n, dt = _timed_loop(run, seconds, max_n=20)
    return {'value': round(Bc * n / dt, 3), 'unit': 'segments/s', 'cores': nt, 'kind': 'port', 'host_cpus': ncpu,
            'sample': "%d train steps ('MLE' forward + autograd backward + clip + Adam) of B=%d with oracle/gvd_oracle.py, "
                      '%.1f s, %d threads' % (n, Bc, dt, nt),
            'reference_in_build_container': _reference_timing() if full else None}


def _attn_fetched_bytes(att_mask_rows, Ft, A, H, group=1):
    """Bytes one attention launch really FETCHES, exactly, from the mask: the kernel skips the projection row and the feature
    row of every region the attention mask removes (csrc/attention.hip: weight exactly 0), per 50-row chunk (a chunk without
    any live row keeps its feature rows: uniform weights).  att_mask_rows: u8 [B, R] (1 = masked), B = samples (for the
    beam-grouped kernel a row is skipped when masked for every beam of the sample: same mask for all beams here)."""
    B, R = att_mask_rows.shape
    chunk = 50
    while chunk > 20 and B * ((R + chunk - 1) // chunk) < 512:        # attention.hip pick_chunk
        chunk = (chunk + 1) // 2
    chunk = max(1, min(chunk, 64, R))
    live = (att_mask_rows == 0)
    pad = (-R) % chunk
    if pad:
        live = torch.cat([live, torch.zeros(B, pad, dtype=torch.bool, device=live.device)], 1)
    per_chunk = live.view(B, -1, chunk).sum(-1)                        # live rows per (sample, chunk)
    rows_in_chunk = torch.full_like(per_chunk, chunk)
    if pad:
        rows_in_chunk[:, -1] = chunk - pad
    p_rows = int(per_chunk.sum())
    f_rows = int(torch.where(per_chunk > 0, per_chunk, rows_in_chunk).sum())
    return 4 * (p_rows * A + f_rows * H) + 4 * B * Ft * (A + H)


def _roofline(attn_ms, attn_n, bytes_per_launch, traffic, kernel, fetched_bytes=None):
    """`frac` / `achieved`: the bytes the launch FETCHES (live rows only, exact from the mask: _attn_fetched_bytes) over the
    HIP-event time; `frac_algorithmic` / `achieved_algorithmic`: SURVEY 8(d)'s all-rows figure B (R + Ft) (A + H) 4 over the
    same time (it counts rows the kernel never reads); `traffic` / `frac_physical`: the PMC bytes of separate counter passes."""
    avg_s = (attn_ms / max(attn_n, 1)) * 1e-3
    alg = bytes_per_launch / avg_s / 1e9 if attn_n else None
    fb = bytes_per_launch if fetched_bytes is None else fetched_bytes
    achieved = fb / avg_s / 1e9 if attn_n else None
    return {'bound': 'hbm', 'kernel': kernel,
            'achieved': None if achieved is None else round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': None if achieved is None else round(achieved / HBM_PEAK_GBS, 4),
            'bytes_per_launch': fb,
            'bytes_basis': ('fetched rows only (rows the attention mask removes are never read), exact from the mask'
                            if fetched_bytes is not None else 'all rows (SURVEY 8d)'),
            'achieved_algorithmic': None if alg is None else round(alg, 1),
            'frac_algorithmic': None if alg is None else round(alg / HBM_PEAK_GBS, 4),
            'algorithmic_bytes_all_rows': bytes_per_launch,
            'traffic': traffic[0], 'traffic_source': traffic[1],
            'hbm_rate_from_traffic_GBs': None if (not attn_n or not traffic[0]) else round(traffic[0] / avg_s / 1e9, 1),
            'frac_physical': None if (not attn_n or not traffic[0]) else round(traffic[0] / avg_s / 1e9 / HBM_PEAK_GBS, 4),
            'avg_launch_us': round(avg_s * 1e6, 2), 'launches_timed': attn_n}


def _lib_srchash():
    """Source hash of the libgvd_hip.so this process loaded (build.py stamps it next to the library)."""
    try:
        with open(os.path.join(ROOT, 'grounded-video-description_amd', 'libgvd_hip.so.srchash')) as f:
            return f.read().strip()
    except OSError:
        return None


def _pmc_traffic(kind, B, Ft, R):
    """HBM bytes per launch of an attention kernel from the PMC counters (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction of
    MI355X_MICROARCH.md), collected by separate `rocprofv3 --pmc` passes (tools/profile_attn.py -> tools/make_attn_traffic.py
    -> profiles/attn_traffic.json), NOT in this run.  The file records the source hash of the library it was measured on: a
    file from another build of the kernels is REFUSED (traffic: null) rather than quoted as if it described this binary."""
    tpath = os.path.join(ROOT, 'profiles', 'attn_traffic.json')
    if not os.path.exists(tpath):
        return None, 'no profiles/attn_traffic.json'
    with open(tpath) as f:
        tj = json.load(f)
    have = _lib_srchash()
    if not have or tj.get('lib_srchash') != have:
        return None, ('profiles/attn_traffic.json was measured on library %s, this is %s: refused (stale)'
                      % (str(tj.get('lib_srchash'))[:12], str(have)[:12]))
    e = tj.get(kind) or {}
    if e.get('batch') == B and e.get('t_attn') == Ft and e.get('regions') == R:
        return e.get('hbm_bytes_per_launch'), ('profiles/attn_traffic.json (rocprofv3 --pmc passes of this workload on this '
                                               'library build, head %s)' % tj.get('head', '?'))
    return None, 'profiles/attn_traffic.json holds no entry for this workload'


def _cpu_baseline_or_pointer(args, world, measure):
    """cpu_baseline is measured on rank 0 at N = 1 only (contract: a bounded sample on the box's host cores, once); an
    N > 1 line points at it instead of spending 15 s of an 8-GPU lease on the host."""
    if args.no_cpu_baseline:
        return None
    if world == 1:
        return measure()
    return {'value': None, 'unit': None, 'cores': None, 'kind': 'port',
            'sample': 'not re-measured at n_gpus = %d: see cpu_baseline of the n_gpus = 1 line of the same command' % world,
            'reference_in_build_container': _reference_timing()}


def _bind_to_gpu_numa_node(local, world):
    """Keep this rank's host threads (launch path, synthetic-input generation, the oracle) on the NUMA node of ITS GPU
    (PCI device -> /sys/bus/pci/devices/<bdf>/numa_node), or - where the topology cannot be read - on an even share of
    the cores.  Returns what was done (goes into the JSON line)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
        node = -1
        pr = torch.cuda.get_device_properties(local)
        if all(hasattr(pr, k) for k in ('pci_domain_id', 'pci_bus_id', 'pci_device_id')):
            bdf = '%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            path = '/sys/bus/pci/devices/%s/numa_node' % bdf
            if os.path.exists(path):
                with open(path) as f:
                    node = int(f.read().strip())
        cpus = None
        if node >= 0:
            cpus = set()
            with open('/sys/devices/system/node/node%d/cpulist' % node) as f:
                for part in f.read().strip().split(','):
                    a, _, b = part.partition('-')
                    cpus.update(range(int(a), int(b or a) + 1))
            cpus &= set(allowed)
            how = 'NUMA node %d of GPU %d' % (node, local)
        if not cpus:
            n = max(1, len(allowed) // world)
            cpus = set(allowed[(local % world) * n:(local % world + 1) * n]) or set(allowed)
            how = 'even share %d/%d of the cores (GPU NUMA node unknown)' % (local % world, world)
        os.sched_setaffinity(0, cpus)
        torch.set_num_threads(max(1, min(len(cpus), (os.cpu_count() or 1) // world)))
        return '%s: %d cpus, %d torch threads' % (how, len(cpus), torch.get_num_threads())
    except (OSError, ValueError, AttributeError, RuntimeError) as e:
        return 'not bound (%s)' % type(e).__name__


def _max_and_per_rank(elapsed, dev, use_dist, gloo):
    """MAX over ranks of the timed region (the contract's clock) + every rank's own time (so a straggler is visible)."""
    if not use_dist:
        return elapsed, [round(elapsed, 6)]
    t = torch.tensor([elapsed], device='cpu' if gloo else dev, dtype=torch.float64)
    allt = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(allt, t)
    per = [round(float(x), 6) for x in allt]
    return max(per), per


def _roofline_mfma(hip, run, steps, label):
    """FLOPs of the pipelined fp32-MFMA GEMM launches (csrc/gemm_pipe.hip) / their HIP-event time, in a dedicated pass of
    `steps` calls of run() OUTSIDE the timed region (the hook adds a one-thread flop-counter kernel per launch).  Rows come
    from the device-side row counts, so the compacted preamble is counted with the rows it really multiplies."""
    prof = hip.GemmProfile(max_pairs=8192)
    with prof:
        for _ in range(steps):
            run()
        ms, n, flops = prof.read()
    if not n or ms <= 0:
        return None
    ach = flops / (ms * 1e-3) / 1e12
    # per shape (live rows, N, K, batch, forward / dX / dW): which products pull the aggregate down
    table = prof.table(MFMA_F32_PEAK_TFS)
    for r in table:
        r['launches'] //= steps
    return {'bound': 'mfma', 'kernel': 'gemm_pipe_kernel (every fp32-MFMA product with >= 256 tiles: %s)' % label,
            'achieved': round(ach, 2), 'peak': MFMA_F32_PEAK_TFS, 'unit': 'TFLOP/s', 'frac': round(ach / MFMA_F32_PEAK_TFS, 4),
            'traffic': None, 'flops_per_step': flops / steps, 'gemm_ms_per_step': round(ms / steps, 3),
            'launches_per_step': n // steps, 'steps_measured': steps, 'per_shape': table}


def section_train_b64(dev, n_steps=4, cpu_seconds=0.0):
    """BASELINE configs[2] inside the default line: one optimisation step of 64 segments (train mode: dropout, BN batch
    statistics), on the weights + inputs of the committed reference case mle_b64_v5000_ft10_trained; `parity` = the
    eval-mode 'MLE' losses of that case against the reference's own (tests/golden), before any update."""
    import numpy as np
    from gvd_amd import att_model, hip, ops, opts, synth, train
    opt = opts.default_opt(vocab_size=5000, t_attn_size=10)
    sd = synth.init_state_dict(opt, seed=5, profile='trained_like')
    model = att_model.TopDownModel(opt)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    a = synth.as_args(synth.trim_to_batch(synth.make_inputs(opt, 64, seed=5, train=True)), dev)
    out = {'batch': 64}
    gpath = os.path.join(GOLDEN_DIR, 'mle_b64_v5000_ft10_trained.npz')
    if os.path.exists(gpath):
        with torch.no_grad():
            got = torch.cat([l.reshape(1) for l in model(*a, 'MLE')]).cpu().numpy()
        want = np.load(gpath)['losses']
        out['parity'] = {'golden': 'tests/golden/mle_b64_v5000_ft10_trained.npz (reference CPU losses lm/att2/grd/cls)',
                         'losses_got': [float(np.float32(x)) for x in got], 'losses_want': [float(np.float32(x)) for x in want],
                         'losses_got_hex': [float(x).hex() for x in got], 'losses_want_hex': [float(x).hex() for x in want],
                         'max_abs_loss_diff': float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max()),
                         'within_1e-4': bool(np.abs(got - want).max() <= 1e-4)}
    model.train()
    tr = train.Trainer(model, opt)
    for _ in range(2):          # warm-up: allocator growth, weight packs, the first step's bucket discovery
        tr.step(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        tr.step(a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_steps
    out.update(ms_per_step=round(1e3 * dt, 3), segments_per_s=round(64 / dt, 1), steps_timed=n_steps)
    out['roofline'] = _roofline_mfma(hip, lambda: tr.step(a), 1, 'forward, dX and dW (K-strided) products of the step')
    del tr, model
    torch.cuda.empty_cache()
    try:
        out['grounding_stream'] = _grounding_stream(ops, dev, 64, opt.seq_length, a[4].shape[1], opt.att_feat_size)
    except Exception as e:          # noqa: BLE001 - a side entry must not take the benchmark line down
        out['grounding_stream'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
    out['compacted_rows'] = _train_compacted(n_steps)
    if cpu_seconds > 0:
        out['cpu_baseline'] = cpu_baseline_train(opt, sd, cpu_seconds, full=False)
    return out


def _grounding_stream(ops, dev, B, M, R, K, reps=20):
    """The grounding product of the training step on its own (model.py:243-280, 469-480: every caption word's visual
    embedding against the segment's R fc7 region features, masked, + class bias + the region-attention logits): the one
    kernel of the step that is a pure STREAM of the [B, T, 100, 2048]-shaped region tensor (M <= 32 words per segment against
    [R, K]: grounder_fwd_kernel of csrc/stream_mm.hip, one launch per step).  HBM roofline entry of its own: algorithmic
    bytes = the region features once + the words + the row-bias read + the mask + the output, over HIP-event time on the
    launching stream; `traffic` = the PMC bytes of this kernel at this shape from profiles/attn_traffic.json when that file was
    measured on this library build.  Next to it the two streams of its backward (d words: one read of the region tensor;
    d regions: one write of its gradient)."""
    g = torch.Generator(device='cpu').manual_seed(0)
    xt = torch.randn(B, M, K, generator=g).to(dev)
    feats = torch.randn(B, R, K, generator=g).to(dev)
    mask = (torch.rand(B, M, R, generator=g) < 0.3).to(torch.uint8).to(dev)
    mbias = torch.randn(B, M, generator=g).to(dev)
    rowbias = torch.randn(B, M, R, generator=g).to(dev)
    dout = torch.randn(B, M, R, generator=g).to(dev)

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    def entry(kernel, us, nbytes, traffic_key=None):
        ach = nbytes / (us * 1e-6) / 1e9
        tr = _pmc_traffic_entry(traffic_key, B, M, R, K) if traffic_key else (None, None)
        return {'kernel': kernel, 'shape': [B, M, R, K], 'bound': 'hbm', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS,
                'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4), 'traffic': tr[0], 'traffic_source': tr[1],
                'frac_physical': None if not tr[0] else round(tr[0] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                'algorithmic_bytes': nbytes, 'avg_launch_us': round(us, 2), 'launches_timed': reps}
    us = timed(lambda: ops.grounder_stream(xt, feats, mask, mbias, rowbias))
    out = entry('grounder_fwd_kernel<8 waves, 4 ring slots> (csrc/stream_mm.hip: ops.grounder forward)', us,
                4 * (B * R * K + B * M * K + 2 * B * M * R + B * M) + B * M * R, 'grounder_fwd')
    dm, _, dmt = ops.masked_copy_rowsum(dout, mask, want_sum=False, want_t=True)
    out['backward_d_words'] = entry('rows_contract_kernel (one read of the region tensor)',
                                    timed(lambda: ops.rows_contract(dm, feats, S_t=dmt)), 4 * (B * R * K + B * R * 32 + B * M * K),
                                    'grounder_d_words')
    out['backward_d_regions'] = entry('rank_update_kernel<3> (one write of the region tensor\'s gradient)',
                                      timed(lambda: ops.rank_update(dm, xt)), 4 * (B * R * K + B * M * R + B * M * K),
                                      'grounder_d_regions')
    return out


def _pmc_traffic_entry(key, B, M, R, K):
    """PMC bytes per launch of a stream_mm kernel from profiles/attn_traffic.json (same staleness rule as _pmc_traffic)."""
    tpath = os.path.join(ROOT, 'profiles', 'attn_traffic.json')
    if not os.path.exists(tpath):
        return None, 'no profiles/attn_traffic.json'
    with open(tpath) as f:
        tj = json.load(f)
    have = _lib_srchash()
    if not have or tj.get('lib_srchash') != have:
        return None, 'profiles/attn_traffic.json was measured on another library build: refused (stale)'
    e = tj.get(key) or {}
    if e.get('shape') == [B, M, R, K]:
        return e.get('hbm_bytes_per_launch'), 'profiles/attn_traffic.json (rocprofv3 --pmc passes on this library build)'
    return None, 'profiles/attn_traffic.json holds no entry for this shape'


def _train_compacted(n_steps):
    """The same step on the compacted training layout (GVD_TRAIN_COMPACT=1, grounded-video-description_amd/train_compact.py:
    per segment its valid proposals + ONE weighted representative of the masked ones the loader zeroed; same losses and
    gradients in eval-mode arithmetic - tests/test_train_compact_cpu.py, test_train_compaction_matches_full_rows).  An opt-in
    (its larger cases have not been through the GPU suite; with live dropout one draw stands for the n masked rows), so it
    is reported NEXT TO the configs[2] number, never as it, and measured by tools/train_compact_bench.py in a SUBPROCESS:
    whatever happens there is recorded and cannot touch this process or its line."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE', 'GROUP_RANK', 'MASTER_ADDR', 'MASTER_PORT',
                        'TORCHELASTIC_RUN_ID', 'GVD_TRAIN_COMPACT')}
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'train_compact_bench.py'), str(n_steps)], env=env,
                           cwd=ROOT, capture_output=True, text=True, timeout=300)
        for line in r.stdout.splitlines():
            if line.startswith('COMPACT_JSON '):
                return json.loads(line[len('COMPACT_JSON '):])
        return {'error': 'rc %d: %s' % (r.returncode, (r.stderr or r.stdout)[-300:])}
    except Exception as e:          # noqa: BLE001 - a side measurement must not take the benchmark line down
        return {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}


def section_beam5_t20_b64(dev, n_steps=10, cpu_seconds=0.0, cpu_threads=None):
    """BASELINE configs[4] inside the default line: beam=5 over 20 frames x 100 regions, 64 segments.  The timed batch IS
    the committed reference case beam5_b64_v5000_ft10_t20 (weights + inputs from seed 22, oracle/cases.py; the reference's
    own beam_search under the harness shim): `parity` compares the ids / attended regions the TIMED region produced - the
    nontemporal attn_partial_group_kernel<5> instantiation, 320 beam rows - with the reference's output."""
    import numpy as np
    from gvd_amd import att_model, hip, ops, opts, synth
    opt = opts.default_opt(vocab_size=5000, t_attn_size=10, num_sampled_frm=20)
    sd = synth.init_state_dict(opt, seed=22, profile='trained_like')
    model = att_model.TopDownModel(opt)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    keys = ('segs_feat', 'ppls', 'num', 'ppls_feat', 'sample_idx', 'pnt_mask')
    out = {'batch': 64, 'beam': 5, 'frames': 20}
    gpath = os.path.join(GOLDEN_DIR, 'beam5_b64_v5000_ft10_t20.npz')
    with torch.no_grad():
        inp = synth.make_inputs(opt, 64, seed=22, train=False)
        d = [inp[k].to(dev) for k in keys]
        for _ in range(2):
            model._sample(*d, {'beam_size': 5})
        timer = hip.KernelTimer(max_pairs=opt.seq_length * n_steps)
        ops.set_kernel_timer(timer)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            seq, lps, att2, _ = model._sample(*d, {'beam_size': 5})
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_steps
        ops.set_kernel_timer(None)
    model.check_kernel_status()
    if os.path.exists(gpath):
        g = np.load(gpath)
        out['parity'] = {'golden': 'tests/golden/beam5_b64_v5000_ft10_t20.npz (reference beam_search under the shim, the TIMED batch)',
                         'compared': 'output of the last timed step',
                         'token_ids_equal': bool((seq.cpu().numpy() == g['seq']).all()),
                         'attended_regions_equal': bool((att2.cpu().numpy() == g['att2']).all()),
                         'max_abs_logprob_diff': float(np.abs(lps.cpu().numpy() - g['seqLogprobs']).max())}
    ms, n = timer.read()
    R = opt.num_sampled_frm * opt.num_prop_per_frm
    out.update(ms_per_step=round(1e3 * dt, 3), captions_per_s=round(64 / dt, 1), steps_timed=n_steps)
    out['roofline'] = _roofline(ms, n, 64 * (R + 10) * (opt.att_hid_size + opt.rnn_size) * 4, _pmc_traffic('beam', 64, 10, R),
                                'attn_partial_group_kernel<5, nontemporal> (one feature stream per sample for its 5 beams)',
                                fetched_bytes=_attn_fetched_bytes(inp['pnt_mask'][:, 1:], 10, opt.att_hid_size, opt.rnn_size))
    del model
    torch.cuda.empty_cache()
    if cpu_seconds > 0:
        out['cpu_baseline'] = cpu_baseline(opt, sd, cpu_seconds, beam=5, threads=cpu_threads)
    return out


def section_ft480_b256(dev, n_steps=3, cpu_seconds=0.0, cpu_threads=None):
    """The reference's DEFAULT frame count (opts.py:50 `--t_attn_size 480`: [B,480,3072] frame features through the two
    frame embeddings, BatchNorm, the 2-layer bi-GRU and a 480-row temporal attention stream) inside the default line:
    greedy decode of 256 segments; `parity` = the committed reference case greedy_b64_v5000_ft480_trained decoded by the
    same code path (2-group persistent GRU, 16 hidden units per workgroup)."""
    import numpy as np
    from gvd_amd import att_model, opts, synth
    from gvd_amd.att_model import attended_region_indices
    opt = opts.default_opt(vocab_size=5000, t_attn_size=480)
    sd = synth.init_state_dict(opt, seed=15, profile='trained_like')
    model = att_model.TopDownModel(opt)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    keys = ('segs_feat', 'ppls', 'num', 'ppls_feat', 'sample_idx', 'pnt_mask')
    out = {'batch': 256, 't_attn_size': 480}
    gpath = os.path.join(GOLDEN_DIR, 'greedy_b64_v5000_ft480_trained.npz')
    with torch.no_grad():
        if os.path.exists(gpath):
            g = np.load(gpath)
            small = synth.make_inputs(opt, 64, seed=15, train=False)
            seq, lps, att2, _ = model._sample(*[small[k].to(dev) for k in keys])
            idx = attended_region_indices(att2, opt.num_sampled_frm, opt.num_prop_per_frm).cpu().numpy()
            out['parity'] = {'golden': 'tests/golden/greedy_b64_v5000_ft480_trained.npz (reference CPU output)',
                             'token_ids_equal': bool((seq.cpu().numpy() == g['seq']).all()),
                             'attended_region_indices_equal': bool((idx == g['att_idx'].astype(idx.dtype)).all())}
            del small, seq, lps, att2
        # the 256-segment batch: four copies of a 64-segment synthetic batch (1.5 GB of frame features are not worth
        # generating on the host for a timing run; the arithmetic does not depend on the values)
        inp = synth.make_inputs(opt, 64, seed=101, train=False)
        d = [torch.cat([inp[k].to(dev)] * 4, 0) for k in keys]
        for _ in range(2):
            model._sample(*d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            model._sample(*d)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_steps
    model.check_kernel_status()
    out.update(ms_per_step=round(1e3 * dt, 3), captions_per_s=round(256 / dt, 1), steps_timed=n_steps)
    del model, d
    torch.cuda.empty_cache()
    if cpu_seconds > 0:
        out['cpu_baseline'] = cpu_baseline(opt, sd, cpu_seconds, threads=cpu_threads)
    return out


def section_files_to_captions(dev, n_seg=1024, B=64):
    """The three separately measured stages COMPOSED (dataloader_anet.py:175-354 -> main.py:314-450): a synthetic split in
    the reference's on-disk layout (.npy region features [10,100,2048] per segment, resnet / bn frame features per video,
    reference-default 480 temporal positions) -> ingest.InferenceIngest (native .npy reader into pinned staging, async H2D,
    zero fill on the device) -> TopDownModel.sample_pipelined (preamble || token loop on two HIP streams) ->
    driver.collect_predictions + the densecap JSON writer.  Wall clock from the first file read to the JSON on disk; next
    to it the same split through the ingest alone and the decode alone (inputs resident), i.e. what each stage would allow.
    1024 segments = 16 batches: the pipeline's fill (the first batch's file reads + upload, ~37 ms at this shape, with nothing
    to hide behind) is 1 / 17 of the run as it is 1 / 270 of the 17 k-segment validation split; the 4-batch split of round 4
    measured mostly that fill."""
    import shutil
    import tempfile
    from gvd_amd import att_model, driver, ingest, opts, synth
    need = n_seg * (1000 * 2048 * 4) + (n_seg // 4 + 1) * 600 * 3072 * 4
    base = None
    for cand in ('/dev/shm', tempfile.gettempdir()):
        try:
            st = os.statvfs(cand)
            if st.f_bavail * st.f_frsize > 1.3 * need:
                base = cand
                break
        except OSError:
            pass
    if base != '/dev/shm' and n_seg > 256:
        # no room in /dev/shm for the 16-batch split (10 GB): do not spend minutes writing it to a disk-backed directory
        # inside a benchmark run - fall back to the 4-batch split
        n_seg = 256
        need = n_seg * (1000 * 2048 * 4) + (n_seg // 4 + 1) * 600 * 3072 * 4
        base = None
        for cand in ('/dev/shm', tempfile.gettempdir()):
            try:
                st = os.statvfs(cand)
                if st.f_bavail * st.f_frsize > 1.3 * need:
                    base = cand
                    break
            except OSError:
                pass
    if base is None:
        return {'skipped': 'no scratch directory with %.1f GB free' % (1.3 * need / 1e9)}
    root = tempfile.mkdtemp(prefix='gvd_split_', dir=base)
    try:
        opt = opts.default_opt(vocab_size=5000, t_attn_size=480)
        opt.id = 'bench'
        t0 = time.perf_counter()
        fr, sr, recs = synth.write_feature_split(root, opt, n_seg, seed=3)
        t_write = time.perf_counter() - t0
        model = att_model.TopDownModel(opt)
        model.load_state_dict(synth.init_state_dict(opt, seed=15, profile='trained_like'))
        model = model.to(dev).eval()
        itow = {str(i): 'w%d' % i for i in range(1, opt.vocab_size)}
        ing = ingest.InferenceIngest(opt, fr, sr, device=dev, max_batch=B)
        out_dir = os.path.join(root, 'results')
        driver.eval_split(model, ing, recs[:2 * B], B, itow, opt, pipelined=True)           # warm-up: page cache, allocator
        torch.cuda.synchronize()
        keys = ('segs_feat', 'ppls', 'num', 'ppls_feat', 'sample_idx', 'pnt_mask')
        per_seg = 1000 * 2048 * 4 + 480 * 3072 * 4 + 1000 * 7 * 4

        def measure(rs, keep_resident):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            preds, _ = driver.eval_split(model, ing, rs, B, itow, opt, out_dir=out_dir, pipelined=True)
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            assert sum(len(v) for v in preds.values()) == len(rs)
            assert os.path.exists(os.path.join(out_dir, 'densecap-validation-bench.json'))
            # the stages alone, same records
            t0 = time.perf_counter()
            held = []
            for chunk, t in ing.batches(rs, B):
                if keep_resident:
                    held.append(tuple(t[k] for k in keys))
            torch.cuda.synchronize()
            t_ing = time.perf_counter() - t0
            r = {'segments': len(rs), 'captions_per_s': round(len(rs) / t_all, 1), 'seconds': round(t_all, 3),
                 'ingest_alone_segments_per_s': round(len(rs) / t_ing, 1),
                 'pcie_GBs_at_this_rate': round(len(rs) / t_all * per_seg / 1e9, 2)}
            if keep_resident:
                with torch.no_grad():
                    model.sample_pipelined(held[:1])
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    model.sample_pipelined(held)
                    torch.cuda.synchronize()
                t_dec = time.perf_counter() - t0
                r['decode_alone_captions_per_s'] = round(len(rs) / t_dec, 1)
            return r
        short = measure(recs[:min(256, len(recs))], True)            # the 4-batch split round 4 measured (mostly pipeline fill)
        full = measure(recs, False) if len(recs) > 256 else dict(short)
        dec = short['decode_alone_captions_per_s']
        for r in (short, full):
            r['fraction_of_the_slower_stage'] = round(r['captions_per_s'] / min(r['ingest_alone_segments_per_s'], dec), 3)
        full.update(batch=B, t_attn_size=480, scratch=base, decode_alone_captions_per_s=dec, dataset_write_seconds=round(t_write, 1),
                    reader_threads=ing.workers, cpu_budget=round(ingest.cpu_budget(), 1),
                    read=('mapped' if ing.read_mode == 1 else 'pread'),
                    short_split_256=short,
                    path="synth.write_feature_split -> ingest.InferenceIngest -> TopDownModel.sample_pipelined -> "
                         "driver.collect_predictions + densecap-<split>-<id>.json")
        return full
    except Exception as e:          # noqa: BLE001 - a side measurement must not take the benchmark line down
        return {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def section_dp_train(args, dev, rank, world, n_steps=6, B=32):
    """BASELINE configs[3] (batch-DP training, 256 segments as 8 x 32; reference main.py:654-658 nn.DataParallel, loss
    semantics main.py:239-255) as a COLLECTIVE-SAFE section of every sample-mode line: every rank runs it (the N = 1 line
    carries the one-GPU point of the same curve, without a collective).  32 segments per GPU, train mode, `n_steps` timed
    optimisation steps between barriers -> whole-job segments/s (MAX over ranks), every rank's seconds, the bucket launch
    timeline of one traced step (when each bucket's all-reduce was issued / when the compute stream got it back, against the
    end of backward), the EXPOSED milliseconds of the gradient exchange (not hidden behind backward) and the collectives
    alone: ring bus bandwidth 2 (N - 1) / N x bytes / t from stream events, to hold against xGMI's ~153 GB/s per link."""
    from gvd_amd import att_model, dist as gdist, opts, synth, train
    use_dist = dist.is_initialized()
    gloo = args.dist_backend == 'gloo'
    opt = opts.default_opt(vocab_size=5000, t_attn_size=10)
    sd = synth.init_state_dict(opt, seed=13, profile='trained_like')
    model = att_model.TopDownModel(opt)
    model.load_state_dict(sd)
    model = model.to(dev).train()
    gdist.broadcast_parameters(model)
    tr = train.Trainer(model, opt)
    a = synth.as_args(synth.trim_to_batch(synth.make_inputs(opt, B, seed=300 + rank, train=True)), dev)
    for _ in range(2):                  # warm-up: allocator growth, weight packs, bucket discovery (step 0) + first hooked step
        tr.step(a)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        losses = tr.step(a)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed, per_rank = _max_and_per_rank(elapsed, dev, use_dist, gloo)
    timeline = exposed = coll = None
    if tr.reducer.active:
        tr.reducer.trace = True
        tr.step(a)                      # one traced step on every rank, outside the timed region
        torch.cuda.synchronize()
        timeline, exposed = tr.reducer.launch_timeline(), tr.reducer.exposed_ms()
        tr.reducer.trace = False
        coll = tr.reducer.measure_allreduce(reps=5)          # the step's collectives alone (every rank)
    out = {'segments_per_gpu': B, 'global_batch': B * world, 'n_gpus': world, 'steps_timed': n_steps,
           'segments_per_s': round(world * B * n_steps / elapsed, 1), 'ms_per_step': round(1e3 * elapsed / n_steps, 3),
           'per_rank_seconds': per_rank, 'losses_last_rank0': [round(float(x), 5) for x in losses],
           'parallelism': ('dp%d: one process per GPU, bucketed gradient all-reduce (%s) launched from autograd hooks in '
                           'gradient-ready order, mean over replicas of per-replica masked means (main.py:239-255)'
                           % (world, 'gloo: code-path run, not a measurement' if gloo else 'RCCL over xGMI'))
                          if world > 1 else 'one GPU: no collective (the N = 1 point of the configs[3] curve)',
           'dp_bucket_launches': timeline, 'allreduce_exposed_ms': exposed, 'allreduce_alone': coll,
           'allreduce_hidden_fraction': (None if not (coll and exposed is not None and coll['ms_per_sweep'] > 0) else
                                         round(max(0.0, 1.0 - exposed / coll['ms_per_sweep']), 3))}
    del tr, model, a
    torch.cuda.empty_cache()
    return out


def bench_train(args, opt, sd, model, B, rank, world, dev):
    """BASELINE configs[2]/[3]: one optimisation step = 'MLE' forward (LM + attention + grounding + cls losses),
    hand-scheduled BPTT, RCCL gradient all-reduce (N>1), clip 0.1, Adam.  Train mode (dropout, BN batch stats)."""
    from gvd_amd import dist as gdist, hip, ops, synth, train
    model.train()
    gdist.broadcast_parameters(model)
    tr = train.Trainer(model, opt)
    inp = synth.trim_to_batch(synth.make_inputs(opt, B, seed=200 + rank, train=True))
    a = synth.as_args(inp, dev)
    for _ in range(args.warmup):
        tr.step(a)
    torch.cuda.synchronize()
    use_dist = dist.is_initialized()
    timer = hip.KernelTimer(max_pairs=opt.seq_length * max(args.steps, 1))
    ops.set_kernel_timer(timer)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = tr.step(a)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ops.set_kernel_timer(None)
    elapsed, per_rank = _max_and_per_rank(elapsed, dev, use_dist, args.dist_backend == 'gloo')
    attn_ms, attn_n = timer.read()
    dp_timeline = None
    if tr.reducer.active:
        # one extra step (all ranks, outside the timed region) with the reducer's launch trace on: at which point of the
        # backward pass each gradient bucket's all-reduce was issued = the window the collective can hide in
        tr.reducer.trace = True
        tr.step(a)
        torch.cuda.synchronize()
        dp_timeline = tr.reducer.launch_timeline()
        tr.reducer.trace = False
    # the kernel that dominates THIS mode: the pipelined fp32-MFMA GEMM (forward, dX, dW products; ~75 % of the step),
    # measured in one extra step after the timed region.  EVERY rank runs that step (it contains the gradient all-reduce:
    # a step on rank 0 alone would leave its peers' collectives unmatched); rank 0 reports its own GEMM launches.
    roof = _roofline_mfma(hip, lambda: tr.step(a), 1, 'forward, dX and dW (K-strided) products of the step')
    if rank == 0:
        R = opt.num_sampled_frm * opt.num_prop_per_frm
        A, H, Ft = opt.att_hid_size, opt.rnn_size, args.t_attn
        out = {
            'metric': "train segments/sec ('MLE' fwd + BPTT + grad all-reduce + clip + Adam)",
            'value': round(world * B * args.steps / elapsed, 2), 'unit': 'segments/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'train step, %d segments/GPU, L=20, 10x100 regions, Ft=%d, V=%d, w_att2=%.2f '
                                   'w_grd=%.2f w_cls=%.2f' % (B, args.t_attn, args.vocab, opt.w_att2, opt.w_grd, opt.w_cls),
                       'batch_per_gpu': B, 'parallelism': 'dp%d (RCCL bucketed grad all-reduce)' % world,
                       'host_binding': args.host_binding},
            'losses_last': [round(float(x), 5) for x in losses],
            'per_rank_seconds': per_rank,
            'dp_bucket_launches': dp_timeline,
            'roofline': roof,
            # the forward attention streaming kernel of the teacher-forced loop (same kernel, same bytes as in decode)
            'roofline_attention': _roofline(attn_ms, attn_n, B * (R + Ft) * (A + H) * 4, (None, None),
                                            'attn_partial_kernel (forward region+temporal attention of the teacher-forced loop)',
                                            fetched_bytes=_attn_fetched_bytes(inp['pnt_mask'][:, 1:], Ft, A, H)),
            'cpu_baseline': _cpu_baseline_or_pointer(args, world, lambda: cpu_baseline_train(opt, sd, args.cpu_seconds))}
        out['library_gemms'] = ops.library_call_count()     # calls that left libgvd_hip.so for a torch library op (GVD_STRICT: 0)
        out['config']['strict'] = bool(ops.STRICT)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and 'RANK' not in os.environ:
        self_launch(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.dist_backend == 'gloo':
        local = local % max(torch.cuda.device_count(), 1)      # (ranks may share a device: code-path runs on a 1-GPU box)
    torch.cuda.set_device(local)
    use_dist = 'RANK' in os.environ        # launched by torch.distributed.run (also for a single rank)
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if args.dist_backend == 'gloo':
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
    if world != args.gpus:
        sys.exit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    dev = torch.device('cuda', local)
    binding = _bind_to_gpu_numa_node(local, world) if world > 1 else None     # (one rank: the whole host is its own)
    args.host_binding = binding

    import gvd_amd  # noqa: F401
    from gvd_amd import att_model, hip, ops, opts, synth
    if args.files > 0:
        if rank == 0:
            print(json.dumps({'metric': 'captions/sec, feature files -> caption JSON (Ft=480, 10x100 regions, greedy)',
                              'unit': 'captions/s', 'n_gpus': 1, 'dtype': 'f32', 'data': 'synthetic files',
                              **section_files_to_captions(dev, n_seg=args.files, B=args.batch or 64)}))
        return
    opt = opts.default_opt(vocab_size=args.vocab, t_attn_size=args.t_attn, num_sampled_frm=args.frames)
    sd = synth.init_state_dict(opt, seed=0, profile='trained_like')
    model = att_model.TopDownModel(opt)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    B = args.batch or ((256 if args.beam == 1 else 64) if args.mode == 'sample' else 64)
    if args.mode == 'train':
        return bench_train(args, opt, sd, model, B, rank, world, dev)
    # each rank: its own shard of seeded segments (rank r uses seed r).  Rank 0 of the default workload therefore holds
    # exactly the inputs + weights of the committed reference case greedy_b256_v5000_ft10_trained (oracle/cases.py).
    keys = ('segs_feat', 'ppls', 'num', 'ppls_feat', 'sample_idx', 'pnt_mask')
    # Rank 0 draws them on the host (the CPU stream the reference case is keyed on), every other rank on its device: eight
    # ranks drawing 2.1 GB of host normals each would spend minutes of a multi-GPU lease before the first kernel.
    inp = synth.make_inputs(opt, B, seed=rank, train=False, device=None if rank == 0 else dev)
    dinp = [inp[k].to(dev) for k in keys]
    golden = None
    if (rank == 0 and B == 256 and args.beam == 1 and args.vocab == 5000 and args.t_attn == 10 and args.frames == 10
            and os.path.exists(GOLDEN_B256)):
        import numpy as np
        golden = np.load(GOLDEN_B256)
    timer = hip.KernelTimer(max_pairs=opt.seq_length * max(args.steps, 1))
    model.kernel_timer = None

    def barrier():
        if use_dist:
            dist.barrier()

    # the timed region goes through the PUBLIC entry point main.py calls (main.py:353-358): model(segs_feat, seq, gt_seq, num,
    # ppls, gt_boxes, mask_boxes, ppls_feat, frm_mask, sample_idx, pnt_mask, 'sample', eval_opt) with the [B] dummies of
    # main.py:353 - it includes the call's one device->host read of the kernel status words
    dummy = torch.zeros(B, dtype=torch.uint8, device=dev)

    def fwd_args(d, beam=None):
        return (d[0], dummy, dummy, d[2], d[1], dummy, dummy, d[3], dummy, d[4], d[5], 'sample',
                {'sample_max': 1, 'beam_size': args.beam if beam is None else beam})

    with torch.no_grad():
        for _ in range(args.warmup):
            model(*fwd_args(dinp))
        torch.cuda.synchronize()
        barrier()
        model.kernel_timer = timer
        ops.set_kernel_timer(timer if args.beam > 1 else None)     # beam: the grouped attention launches of beam.py
        timer.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if args.h2d:
            # host -> device over PCIe every step: pinned staging, copy stream, two device buffer sets
            host = [t.cpu().pin_memory() for t in dinp]
            dev_sets = [[torch.empty_like(t) for t in dinp] for _ in range(2)]
            copy_stream = torch.cuda.Stream()
            ready = [torch.cuda.Event() for _ in range(2)]
            done = [torch.cuda.Event() for _ in range(2)]
            comp = torch.cuda.current_stream()

            def start_copy(k):
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(done[k])          # the set's previous consumer finished
                    for d, h in zip(dev_sets[k], host):
                        d.copy_(h, non_blocking=True)
                    ready[k].record(copy_stream)
            for e in done:
                e.record(comp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            start_copy(0)
            for i in range(args.steps):
                k = i & 1
                if i + 1 < args.steps:
                    start_copy(k ^ 1)
                comp.wait_event(ready[k])
                seq, att2, sim = model(*fwd_args(dev_sets[k]))
                done[k].record(comp)
        elif not args.overlap:
            for _ in range(args.steps):
                seq, att2, sim = model(*fwd_args(dinp))
        else:
            # all K steps are enqueued on two streams (preamble | token loop) and fully completed before the clock stops
            outs = model.sample_pipelined([dinp] * args.steps)
            seq, _lps, att2, sim = outs[-1]
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        lps = None
        if golden is not None:           # (forward() does not return the log-probs: one more, untimed, call for the parity block)
            lps = model._sample(*dinp)[1]
    ops.set_kernel_timer(None)
    model.check_kernel_status()          # outside the timed region: no persistent-kernel barrier timed out
    elapsed, per_rank = _max_and_per_rank(elapsed, dev, use_dist, args.dist_backend == 'gloo')
    attn_ms, attn_n = timer.read()

    if rank == 0:
        R = opt.num_sampled_frm * opt.num_prop_per_frm
        A, H, Ft = opt.att_hid_size, opt.rnn_size, args.t_attn
        # algorithmic bytes.  Beam: the grouped kernel streams a SAMPLE's features once for
        # its K beams, so the algorithmic bytes per launch are per sample, not per beam row.
        bytes_per_launch = B * (R + Ft) * (A + H) * 4
        out = {
            'metric': 'captions/sec (seq_len=20, %dx100 regions), %s' % (args.frames, 'greedy decode' if args.beam == 1
                                                                         else 'beam-search decode (beam=%d)' % args.beam),
            'value': round(world * B * args.steps / elapsed, 2),
            'unit': 'captions/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic', 'per_rank_seconds': per_rank,
            'library_gemms': None,        # filled in at the end: calls that left libgvd_hip.so for a torch library op
            'config': {'workload': "%s forward(..., 'sample') (preamble + 20-token loop), %d segments/GPU/step, L=20, "
                                   "T x P = %d x 100 regions [B,%d,2048] fc6 + [B,%d,3072] frame feats, V=%d, "
                                   "obj_interact on; random-init weights (trained_like profile)"
                                   % ('greedy' if args.beam == 1 else 'beam=%d' % args.beam, B, args.frames, R, Ft, args.vocab),
                       'batch_per_gpu': B, 'parallelism': 'batch-sharded replicas x%d (no data-path collective)' % world,
                       'overlap': 'preamble(i+1) || token-loop(i) on 2 HIP streams' if args.overlap else 'off (steps run serially)',
                       'inputs': 'copied from pinned host memory every step (PCIe-inclusive)' if args.h2d else 'resident in HBM'},
            'roofline': _roofline(attn_ms, attn_n, bytes_per_launch, _pmc_traffic('greedy' if args.beam == 1 else 'beam', B, Ft, R),
                                  'attn_partial_kernel (region+temporal additive attention)' if args.beam == 1 else
                                  'attn_partial_group_kernel<%d> (one feature stream per sample for its %d beams)'
                                  % (args.beam, args.beam),
                                  fetched_bytes=_attn_fetched_bytes(inp['pnt_mask'][:, 1:], Ft, A, H)),
        }
        out['config']['host_binding'] = binding
        if not args.h2d and not args.overlap:
            # the kernel that takes most of the step's GPU time is not the attention stream but the fp32-MFMA GEMM of the
            # per-segment preamble: its own roofline block, from two extra steps after the timed region (rank 0's launches;
            # the decode path has no collective, the other ranks wait at the closing barrier)
            model.kernel_timer = None
            with torch.no_grad():
                out['roofline_mfma'] = _roofline_mfma(
                    hip, lambda: model._sample(*dinp, {'beam_size': args.beam}), 2,
                    'fc7, class similarity, pool_embed, encoder q|k|v / wo / feed-forward, ctx2pool of the preamble')
        if golden is not None:
            # the timed run decoded the committed reference case: compare with the reference's own output
            from gvd_amd.att_model import attended_region_indices
            ids_ok = bool((seq.cpu().numpy() == golden['seq']).all())
            idx = attended_region_indices(att2, opt.num_sampled_frm, opt.num_prop_per_frm).cpu().numpy()
            idx_ok = bool((idx == golden['att_idx'].astype(idx.dtype)).all())
            out['parity'] = {'golden': 'tests/golden/greedy_b256_v5000_ft10_trained.npz (reference CPU output, '
                                       'oracle/make_golden.py)', 'token_ids_equal': ids_ok,
                             'attended_region_indices_equal': idx_ok,
                             'max_abs_logprob_diff': float(abs(lps.cpu().numpy() - golden['seqLogprobs']).max()),
                             'note': 'ids / regions: the output of the TIMED forward() calls; log-probs: one more untimed call'}
        if args.beam == 1 and B != 4 and not args.h2d:
            # BASELINE configs[1] shape (batch_size=4 eval) next to the headline batch: the latency-bound case (rank 0)
            model.kernel_timer = None
            # through the PUBLIC entry point main.py calls, model(segs_feat, seq, gt_seq, num, ppls, gt_boxes, mask_boxes,
            # ppls_feat, frm_mask, sample_idx, pnt_mask, 'sample', eval_opt) with the [B] dummies of main.py:353 - it
            # includes the call's one device->host read of the kernel status words
            s4 = {k: t[:4].contiguous() for k, t in zip(keys, dinp)}
            dummy4 = dummy[:4]
            fwd = (s4['segs_feat'], dummy4, dummy4, s4['num'], s4['ppls'], dummy4, dummy4, s4['ppls_feat'], dummy4,
                   s4['sample_idx'], s4['pnt_mask'], 'sample', {'sample_max': 1, 'beam_size': 1})
            del seq, lps, att2, sim
            torch.cuda.empty_cache()     # a separate measurement: do not carve 32 MB tensors out of cached multi-GB blocks
            with torch.no_grad():
                for _ in range(3):
                    model(*fwd)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(20):
                    model(*fwd)
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / 20
            out['config']['configs1_b4'] = {'batch': 4, 'ms_per_call': round(1e3 * dt, 3), 'captions_per_s': round(4 / dt, 1),
                                            'calls_timed': 20, 'entry_point': "forward(..., 'sample', eval_opt)"}
        default_workload = (world == 1 and args.beam == 1 and B == 256 and not args.h2d and not args.overlap
                            and args.vocab == 5000 and args.t_attn == 10 and args.frames == 10)
        if default_workload and not args.no_sections:
            # BASELINE configs[2] and configs[4] as short sections of the default line (their full lines:
            # `--mode train`, `--beam 5 --frames 20 --batch 64`)
            del model, dinp
            torch.cuda.empty_cache()
            # (each with its own short cpu_baseline: the oracle on the same shapes, ~5 s, at the headline's thread count)
            out['cpu_baseline'] = _cpu_baseline_or_pointer(args, world, lambda: cpu_baseline(opt, sd, args.cpu_seconds))
            cs = 0.0 if args.no_cpu_baseline else min(5.0, args.cpu_seconds)
            ct = (out['cpu_baseline'] or {}).get('cores')
            out['config']['configs2_train_b64'] = section_train_b64(dev, cpu_seconds=cs)
            out['config']['configs4_beam5_t20_b64'] = section_beam5_t20_b64(dev, cpu_seconds=cs, cpu_threads=ct)
            out['config']['ft480_b256'] = section_ft480_b256(dev, cpu_seconds=cs, cpu_threads=ct)
            out['config']['files_to_captions_ft480'] = section_files_to_captions(dev)
        else:
            out['cpu_baseline'] = _cpu_baseline_or_pointer(args, world,
                                                           lambda: cpu_baseline(opt, sd, args.cpu_seconds, beam=args.beam))
    # configs[3] section: COLLECTIVE - every rank runs it (rank 0 after its single-rank extra passes above; the others wait in
    # the section's first collective).  It carries the RCCL evidence of an N > 1 run: all-reduce bus bandwidth, exposed ms.
    dp = None
    if args.beam == 1 and not args.h2d and not args.overlap and not args.no_dp_section:
        dinp = model = None             # (rank 0 of the default workload dropped them before its sections already)
        torch.cuda.empty_cache()
        try:
            dp = section_dp_train(args, dev, rank, world)
        except Exception as e:          # noqa: BLE001 - reported, not fatal (but never swallowed on a subset of ranks: a
            if use_dist and world > 1:  # collective section that fails on one rank must take the job down, not hang it)
                raise
            dp = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
    if rank == 0:
        out['config']['configs3_dp_train'] = dp
        out['library_gemms'] = ops.library_call_count()
        out['config']['strict'] = bool(ops.STRICT)
        print(json.dumps(out))
    barrier()                    # (rank 0's extra measurement passes are over before any rank tears the group down)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
