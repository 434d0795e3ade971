"""Where does the idle quarter of the matrix pipe go in flash_attn_pad_kernel?  Times the kernel at B=256 with parts
removed (GVD_FLASH_ABLATE: results are WRONG, the MFMA work is unchanged): 1 = no online softmax, 2 = no K/V staging /
tile barrier, 3 = both.  One child process per mode (the knob is read once)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get('FA_CHILD'):
    import torch
    import gvd_amd  # noqa: F401
    from gvd_amd import ops
    B, R = 256, 1000
    qkv = torch.randn(B, R, 18 * ops.HEAD_PAD, device='cuda') * 0.3
    for _ in range(2):
        ops.flash_attn_padded(qkv, 6, 1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.flash_attn_padded(qkv, 6, 1.0)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = B * 6 * 2 * 2 * R * R * 176
    print('ablate=%s skew=%s: %.3f ms  (%.1f TF/s incl. pads)' % (os.environ.get('GVD_FLASH_ABLATE', '0'), os.environ.get('GVD_FLASH_SKEW', '1'), ms, fl / ms / 1e9), flush=True)
else:
    for abl in ('0', '1', '2', '3'):
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, FA_CHILD='1', GVD_FLASH_ABLATE=abl), check=False)
