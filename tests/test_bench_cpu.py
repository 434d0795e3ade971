"""CPU: bench.py's launcher contract.  `python bench.py --gpus N` with N above the visible device count must fail with
a clear message instead of dying inside a rendezvous (the driver launches N > 1 itself under torch.distributed.run; a
user calling the script directly gets the self-launch path)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_reports_missing_devices():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    env['HIP_VISIBLE_DEVICES'] = ''
    env['CUDA_VISIBLE_DEVICES'] = ''
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8'], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0
    msg = p.stderr + p.stdout
    assert '--gpus 8 requested' in msg and 'GPU(s)' in msg, msg[-500:]


def test_world_size_mismatch_is_reported():
    """Launched by a torch.distributed launcher with a WORLD_SIZE that contradicts --gpus: refuse before touching a GPU."""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    assert "if world != args.gpus" in src and "sys.exit('bench.py: --gpus %d but WORLD_SIZE=%d'" in src


def test_attention_fetched_bytes_against_a_row_by_row_count():
    """bench._attn_fetched_bytes (roofline.frac's byte count: what the attention kernel fetches, exact from the mask) against a
    literal restatement of the kernel's rule (csrc/attention.hip): per 50-row chunk of a sample the projection rows of its live
    regions, the feature rows of its live regions - or of ALL its rows when the chunk has no live one - plus the temporal side."""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    g = torch.Generator().manual_seed(0)
    for B, R, Ft, p in ((256, 1000, 10, 0.2), (64, 2000, 10, 0.5), (7, 130, 480, 0.97)):
        m = (torch.rand(B, R, generator=g) < p).to(torch.uint8)
        m[0] = 1                                           # a fully masked sample: every chunk keeps its feature rows
        chunk = 50
        while chunk > 20 and B * ((R + chunk - 1) // chunk) < 512:
            chunk = (chunk + 1) // 2
        chunk = max(1, min(chunk, 64, R))
        A, H = 512, 1024
        want = 4 * B * Ft * (A + H)
        for b in range(B):
            for c0 in range(0, R, chunk):
                rows = min(chunk, R - c0)
                live = int((m[b, c0:c0 + rows] == 0).sum())
                want += 4 * live * A + 4 * (live if live else rows) * H
        assert bench._attn_fetched_bytes(m, Ft, A, H) == want
