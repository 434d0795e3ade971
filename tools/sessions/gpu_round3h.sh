#!/bin/bash
# Round-3 session H: split-K for the few-tile products of the B=4 preamble (A/B of the thresholds), raw H2D rate, ingest at Ft=10
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "split_k or gemm" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -p no:cacheprovider -k "b4 or edge or persistent" 2>&1 | tail -3
for cfg in "0:208:640" "1:208:640" "1:256:640" "1:256:800" "1:256:1100"; do
  IFS=: read on mt tot <<< "$cfg"
  GVD_SPLIT_K=$on GVD_SPLIT_K_MAXTILES=$mt GVD_SPLIT_K_MAXTOTAL=$tot timeout 300 python bench.py --batch 4 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('split_k=$cfg', j['ms_per_step'], j['value'])"
done | tee $O/b4_splitk_ab_3h.log
python - <<'PY' 2>&1 | grep -v Warning | tee $O/h2d_rate_3h.log
import torch, time
for mb in (64, 512, 2048):
    h = torch.empty(mb << 18, dtype=torch.float32).pin_memory()
    d = torch.empty_like(h, device='cuda')
    d.copy_(h, non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print('pinned -> device, %4d MB per copy: %.1f GB/s' % (mb, mb / 1024 / dt))
PY
