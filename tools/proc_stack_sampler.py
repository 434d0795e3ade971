"""Sample state / wchan / kernel stack of every task of a process for a while (needs root for /proc/<pid>/task/*/stack).
    python tools/proc_stack_sampler.py <pid> <seconds> <out.txt>
One line per (pass, task): pass index, comm, state, wchan, the top frames of the kernel stack."""
import os
import sys
import time

pid, secs, out = int(sys.argv[1]), float(sys.argv[2]), sys.argv[3]
t_end = time.time() + secs
rows = []
i = 0
while time.time() < t_end:
    try:
        tids = os.listdir('/proc/%d/task' % pid)
    except OSError:
        break
    for t in tids:
        b = '/proc/%d/task/%s/' % (pid, t)
        try:
            st = open(b + 'stat').read()
            state = st[st.rindex(')') + 2]
            comm = st[st.index('(') + 1:st.rindex(')')]
            if state == 'S':
                w = open(b + 'wchan').read()
                if w.startswith('futex') or 'poll' in w or w.startswith('hrtimer') or w.startswith('do_wait') or w.startswith('pipe'):
                    continue
            else:
                w = open(b + 'wchan').read()
            try:
                stack = [l.split('] ')[-1].split('+')[0] for l in open(b + 'stack').read().splitlines()[:6]]
            except OSError:
                stack = ['(no access)']
            rows.append('%d %s %s %s %s' % (i, comm, state, w, '<'.join(stack)))
        except OSError:
            pass
    i += 1
with open(out, 'w') as f:
    f.write('\n'.join(rows) + '\n')
    f.write('# passes %d\n' % i)
