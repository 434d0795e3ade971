#!/bin/bash
# Round-3 session Q: per-dispatch trace of one B=4 'sample' call
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_b4t
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_b4t -o p -- python $R/bench.py --batch 4 --steps 3 --warmup 3 --no-cpu-baseline > $O/prof_b4t.log 2>&1; echo "rc=$?"
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_b4t/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# last call = from the last 'compact_index_kernel' to the end
idx = [i for i, r in enumerate(rows) if 'compact_index' in r['Kernel_Name']]
a = idx[-1] - 8
seq = rows[a:]
t0 = int(seq[0]['Start_Timestamp'])
prev_end = t0
out = open('/root/repo/gpurun_out/b4_dispatch_trace_q.md', 'w') if False else None
lines = []
for r in seq:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]
    lines.append('%8.1f  +%5.1f gap  %7.1f us  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, name))
    prev_end = e
import os
open(os.environ.get('GRAFT_REPO_ROOT', '/root/repo') + '/gpurun_out/b4_dispatch_trace_q.txt', 'w').write('\n'.join(lines))
print(len(seq), 'dispatches; span %.1f us; kernel time %.1f us' % ((prev_end - t0) / 1e3, sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seq) / 1e3))
PY
