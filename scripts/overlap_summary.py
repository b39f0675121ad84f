"""How much of the timed region do kernels of two hardware queues share?  usage: overlap_summary.py <kernel_trace.csv> <out.txt> <header>
Reads a rocprofv3 --kernel-trace CSV (one row per dispatch with queue id, start, end), takes the middle half of the dispatches (inside
the timed steps when the run has many of them), and reports per queue: dispatches, busy time; and the time in which kernels of >= 2 queues run together."""
import csv, sys, glob
from collections import defaultdict

paths = glob.glob(sys.argv[1], recursive=True)
rows = []
for p in paths:
    with open(p) as f:
        for r in csv.DictReader(f):
            k = {a.lower(): b for a, b in r.items()}
            try:
                rows.append((int(k['start_timestamp']), int(k['end_timestamp']), k.get('queue_id', '?'), k.get('kernel_name', '?')))
            except (KeyError, ValueError):
                pass
rows.sort()
rows = rows[int(len(rows) * 0.25):int(len(rows) * 0.75)]          # with --steps 200 this stretch lies inside the timed steps
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = []
per_q = defaultdict(lambda: [0, 0])
for s, e, q, name in rows:
    ev.append((s, 1, q)); ev.append((e, -1, q))
    per_q[q][0] += 1; per_q[q][1] += e - s
ev.sort()
active = defaultdict(int)
last = t0; any_t = 0; multi_q = 0; multi_k = 0
for t, d, q in ev:
    nq = sum(1 for v in active.values() if v > 0); nk = sum(active.values())
    if nk > 0: any_t += t - last
    if nq > 1: multi_q += t - last
    if nk > 1: multi_k += t - last
    active[q] += d; last = t
wall = t1 - t0
with open(sys.argv[2], 'w') as f:
    f.write('# %s\n' % sys.argv[3])
    f.write('# window of the trace: %.3f ms, %d dispatches on %d hardware queues\n' % (wall / 1e6, len(rows), len(per_q)))
    f.write('chip busy (>= 1 kernel running)          %.3f ms  %.1f %% of the window\n' % (any_t / 1e6, 100.0 * any_t / wall))
    f.write('>= 2 kernels running                     %.3f ms  %.1f %%\n' % (multi_k / 1e6, 100.0 * multi_k / wall))
    f.write('kernels of >= 2 queues running together  %.3f ms  %.1f %%\n' % (multi_q / 1e6, 100.0 * multi_q / wall))
    for q, (n, busy) in sorted(per_q.items(), key=lambda kv: -kv[1][1]):
        f.write('queue %-6s %6d dispatches  sum of kernel durations %.3f ms (%.1f %% of the window)\n' % (q, n, busy / 1e6, 100.0 * busy / wall))
print(open(sys.argv[2]).read())
