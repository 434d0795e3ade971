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
