"""Summarise rocprofv3 output directories into small text/JSON files for profiles/.

  python tools/parse_rocprof.py stats <dir> <out.md> [title]     # --kernel-trace --stats run
  python tools/parse_rocprof.py pmc <dir> <kernel-substr> <out.json> <COUNTER> [<COUNTER>...]
  python tools/parse_rocprof.py trace <dir> <out.md> [title]     # per (kernel, grid) groups + idle gaps of the timeline
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, pattern):
    return sorted(glob.glob(os.path.join(d, '**', pattern), recursive=True))


def stats(d, out, title):
    files = find(d, '*kernel_stats.csv')
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    if not rows:   # fall back to the raw trace
        agg = defaultdict(lambda: [0, 0.0])
        for f in find(d, '*kernel_trace.csv'):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    k = r.get('Kernel_Name') or r.get('Name')
                    dur = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
                    agg[k][0] += 1
                    agg[k][1] += dur
        rows = [{'Name': k, 'Calls': v[0], 'TotalDurationNs': v[1], 'AverageNs': v[1] / v[0]} for k, v in agg.items()]
    tot = sum(float(r['TotalDurationNs']) for r in rows) or 1.0
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    with open(out, 'w') as fh:
        fh.write('# %s\n\nrocprofv3 --kernel-trace --stats; total GPU kernel time %.3f ms; top kernels:\n\n' % (title, tot / 1e6))
        fh.write('| kernel | calls | total ms | avg us | %% |\n|---|---|---|---|---|\n')
        for r in rows[:int(os.environ.get('GVD_STATS_ROWS', '40'))]:
            fh.write('| `%s` | %s | %.3f | %.2f | %.1f |\n' % (r['Name'][:110], r['Calls'], float(r['TotalDurationNs']) / 1e6,
                                                           float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
    print(open(out).read()[:3000])


def pmc(d, substr, out, counters):
    vals = defaultdict(list)
    for f in find(d, '*counter_collection.csv'):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if substr in r.get('Kernel_Name', '') and r.get('Counter_Name') in counters:
                    vals[r['Counter_Name']].append(float(r['Counter_Value']))
    res = {c: {'n': len(v), 'mean': (sum(v) / len(v) if v else None), 'min': min(v) if v else None,
               'max': max(v) if v else None} for c, v in vals.items()}
    with open(out, 'w') as fh:
        json.dump({'kernel': substr, 'counters': res}, fh, indent=1)
    print(json.dumps(res))


def _short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return name.split('(')[0][:70]


def trace(d, out, title):
    """Per (kernel, grid, workgroup) groups of a --kernel-trace run, and how much of the span the GPU sat idle between
    dispatches (a launch-bound stretch shows up as gap time)."""
    ev = []
    for f in find(d, '*kernel_trace.csv'):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                g = tuple(int(r.get(k, 0) or 0) for k in ('Grid_Size_X', 'Grid_Size_Y', 'Grid_Size_Z'))
                w = int(r.get('Workgroup_Size_X', 0) or 0)
                ev.append((float(r['Start_Timestamp']), float(r['End_Timestamp']), _short(r.get('Kernel_Name') or r.get('Name')), g, w))
    ev.sort()
    if not ev:
        print('no kernel trace under', d)
        return
    agg = defaultdict(lambda: [0, 0.0])
    busy, gaps, gap_n, last_end = 0.0, 0.0, 0, ev[0][0]
    small_n, small_t = 0, 0.0
    for s0, e0, k, g, w in ev:
        agg[(k, g, w)][0] += 1
        agg[(k, g, w)][1] += e0 - s0
        busy += e0 - s0
        if e0 - s0 < 10e3:
            small_n += 1
            small_t += e0 - s0
        if s0 > last_end:
            gaps += s0 - last_end
            gap_n += 1
        last_end = max(last_end, e0)
    span = last_end - ev[0][0]
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(out, 'w') as fh:
        fh.write('# %s\n\n%d dispatches; span %.3f ms; kernel time %.3f ms; idle between dispatches %.3f ms (%d gaps); '
                 'dispatches under 10 us: %d = %.3f ms\n\n' % (title, len(ev), span / 1e6, busy / 1e6, gaps / 1e6, gap_n,
                                                                small_n, small_t / 1e6))
        fh.write('| kernel | grid (threads) | wg | calls | total ms | avg us |\n|---|---|---|---|---|---|\n')
        for (k, g, w), (n, t) in rows[:400]:
            fh.write('| `%s` | %s | %d | %d | %.3f | %.2f |\n' % (k, 'x'.join(str(x) for x in g), w, n, t / 1e6, t / n / 1e3))
    print(open(out).read()[:6000])


def timeline(d, out, min_us):
    """Dispatches of at least `min_us` in issue order with start offsets (ms) and durations: shows how the same kernel's
    duration depends on what ran before it (clock / power-state ramps after light phases)."""
    ev = []
    for f in find(d, '*kernel_trace.csv'):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                g = int(r.get('Grid_Size_X', 0) or 0) * max(1, int(r.get('Grid_Size_Y', 1) or 1))
                ev.append((float(r['Start_Timestamp']), float(r['End_Timestamp']), _short(r.get('Kernel_Name') or r.get('Name')), g))
    ev.sort()
    t0 = ev[0][0]
    with open(out, 'w') as fh:
        fh.write('| start ms | dur us | grid threads | kernel |\n|---|---|---|---|\n')
        for s0, e0, k, g in ev:
            if (e0 - s0) / 1e3 >= min_us:
                fh.write('| %.3f | %.1f | %d | `%s` |\n' % ((s0 - t0) / 1e6, (e0 - s0) / 1e3, g, k))
    print(open(out).read()[-6000:])


if __name__ == '__main__':
    if sys.argv[1] == 'timeline':
        timeline(sys.argv[2], sys.argv[3], float(sys.argv[4]) if len(sys.argv) > 4 else 300.0)
    elif sys.argv[1] == 'trace':
        trace(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else 'kernel trace')
    elif sys.argv[1] == 'stats':
        stats(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else 'kernel stats')
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5:])
