#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
echo "numa_balancing: $(cat /proc/sys/kernel/numa_balancing 2>/dev/null)"; cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/shmem_enabled 2>/dev/null
snap() { grep -E "^(numa_|pgmigrate|thp_fault|thp_split|pgfault|pgmajfault|compact_stall|nr_tlb|tlb_|allocstall)" /proc/vmstat | tr '\n' ' '; echo; }
snap > /tmp/v0
timeout 300 python tools/files_timeline.py 512 64 3 2 2>&1 | grep "captions/s"
snap > /tmp/v1
python - <<'PY'
a = dict(zip(*[iter(open('/tmp/v0').read().split())] * 2)); b = dict(zip(*[iter(open('/tmp/v1').read().split())] * 2))
print({k: int(b[k]) - int(a[k]) for k in a if int(b[k]) != int(a[k])})
PY
echo "== numa_balancing off for one run"
echo 0 > /proc/sys/kernel/numa_balancing 2>/dev/null; cat /proc/sys/kernel/numa_balancing
timeout 300 python tools/files_timeline.py 512 64 3 2 2>&1 | grep "captions/s\|^  [2-7] " | cut -c1-100
echo 1 > /proc/sys/kernel/numa_balancing 2>/dev/null
echo "== interleave"
numactl --interleave=all python tools/files_timeline.py 512 64 3 2 2>&1 | grep "captions/s\|^  [2-7] " | cut -c1-100
