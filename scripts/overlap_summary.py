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
# the MFMA-bound launches (implicit-GEMM kernels of at least 40 us): how long is at least one of them on the chip, how long two of them together,
# and how long does NOTHING run -- what two lanes can and cannot hide
big = sorted((s_, e_) for s_, e_, q_, n_ in rows if ('ry_igemm' in n_ or 'ry_wino' in n_) and e_ - s_ >= 40000)
bev = sorted([(s_, 1) for s_, _ in big] + [(e_, -1) for _, e_ in big])
nb = 0; lastb = t0; big1 = 0; big2 = 0
for t, d in bev:
    if nb > 0: big1 += t - lastb
    if nb > 1: big2 += t - lastb
    nb += d; lastb = t
gaps = []
cur_end = None
for s_, e_, q_, n_ in rows:
    if cur_end is not None and s_ > cur_end:
        gaps.append(s_ - cur_end)
    cur_end = e_ if cur_end is None else max(cur_end, e_)
gaps.sort(reverse=True)
with open(sys.argv[2], 'w') as f:
    f.write('# %s\n' % sys.argv[3])
    f.write('# window of the trace: %.3f ms, %d dispatches on %d hardware queues\n' % (wall / 1e6, len(rows), len(per_q)))
    f.write('chip busy (>= 1 kernel running)          %.3f ms  %.1f %% of the window\n' % (any_t / 1e6, 100.0 * any_t / wall))
    f.write('>= 2 kernels running                     %.3f ms  %.1f %%\n' % (multi_k / 1e6, 100.0 * multi_k / wall))
    f.write('kernels of >= 2 queues running together  %.3f ms  %.1f %%\n' % (multi_q / 1e6, 100.0 * multi_q / wall))
    f.write('nothing running                          %.3f ms  %.1f %%  (%d gaps; median %.1f us, the 10 longest %s us)\n' % (
        (wall - any_t) / 1e6, 100.0 * (wall - any_t) / wall, len(gaps), (gaps[len(gaps) // 2] / 1e3 if gaps else 0.0),
        ' '.join('%.0f' % (g / 1e3) for g in gaps[:10])))
    f.write('>= 1 MFMA-bound launch (igemm | wino >= 40 us)   %.3f ms  %.1f %%   (%d such launches, sum of their durations %.3f ms)\n' % (
        big1 / 1e6, 100.0 * big1 / wall, len(big), sum(e_ - s_ for s_, e_ in big) / 1e6))
    f.write('>= 2 MFMA-bound launches together         %.3f ms  %.1f %%\n' % (big2 / 1e6, 100.0 * big2 / wall))
    for q, (n, busy) in sorted(per_q.items(), key=lambda kv: -kv[1][1]):
        f.write('queue %-6s %6d dispatches  sum of kernel durations %.3f ms (%.1f %% of the window)\n' % (q, n, busy / 1e6, 100.0 * busy / wall))
print(open(sys.argv[2]).read())
