"""Where is no MFMA-bound launch on the chip?  usage: lane_timeline.py <kernel_trace.csv glob> <out.txt> <header>
Reads a rocprofv3 --kernel-trace CSV of the two-lane bench, takes the middle half of the dispatches, labels every stage-2 dispatch of a hardware queue with
its position in the window's launch sequence (0 = ry_pad_min_rows), and splits the time in which no implicit-GEMM launch of >= 40 us is running by WHAT the two
stage-2 queues are running instead (a pair of sequence labels, or idle).  Also: per sequence position, the average duration and the average wait in front of it.
CAUTION: under the tracer the host needs longer to enqueue a window than the chip needs to run it (round 5: 623 us of queue idle in front of every window's first
kernel, 1.154 instead of 1.09 ms per window) -- the idle shares describe the traced run, the per-position durations are the reliable part."""
import csv, glob, sys
from collections import defaultdict

rows = []
for p in glob.glob(sys.argv[1], recursive=True):
    with open(p) as f:
        for r in csv.DictReader(f):
            k = {a.lower(): b for a, b in r.items()}
            try:
                rows.append((int(k['start_timestamp']), int(k['end_timestamp']), k.get('queue_id', '?'), k.get('kernel_name', '?')))
            except (KeyError, ValueError):
                pass
rows.sort()
rows = rows[int(len(rows) * 0.25):int(len(rows) * 0.75)]
short = lambda n: n.replace('void ', '').split('(')[0].replace(' ', '')
s2q = sorted({q for _, _, q, n in rows if 'ry_pad_min_rows' in n})
lab = {}                                    # (queue, start) -> label
per_pos = defaultdict(lambda: [0, 0.0, 0.0, ''])
for q in s2q:
    pos = None; prev_end = None
    for s, e, qq, n in rows:
        if qq != q:
            continue
        if 'ry_pad_min_rows' in n:
            pos = 0
        elif pos is not None:
            pos += 1
        if pos is None:
            continue
        lab[(q, s)] = '%02d:%s' % (pos, short(n))
        st = per_pos[pos]; st[0] += 1; st[1] += (e - s) / 1e3; st[3] = short(n)
        if prev_end is not None:
            st[2] += max(0, s - prev_end) / 1e3
        prev_end = e
big = [(s, e) for s, e, q, n in rows if ('ry_igemm' in n or 'ry_wino' in n) and e - s >= 40000]
ev = []
for s, e in big:
    ev.append((s, 0, 'B', 1)); ev.append((e, 0, 'B', -1))
for s, e, q, n in rows:
    if q in s2q and (q, s) in lab:
        ev.append((s, 1, (q, lab[(q, s)]), 1)); ev.append((e, 1, (q, lab[(q, s)]), -1))
ev.sort(key=lambda x: (x[0], x[3]))
nb = 0; cur = {q: None for q in s2q}; last = ev[0][0]
acc = defaultdict(float)
t_none = 0.0
for t, kind, what, d in ev:
    if nb == 0:
        key = tuple(sorted((cur[q] or 'idle') for q in s2q))
        acc[key] += (t - last) / 1e3; t_none += (t - last) / 1e3
    if kind == 0:
        nb += d
    else:
        q, l = what
        cur[q] = l if d > 0 else (None if cur[q] == l else cur[q])
    last = t
wall = (rows[-1][1] - rows[0][0]) / 1e3
nwin = per_pos[0][0]
with open(sys.argv[2], 'w') as f:
    f.write('# %s\n' % sys.argv[3])
    f.write('# %d windows in %.1f ms; no MFMA-bound launch (igemm | wino >= 40 us) on the chip for %.1f ms = %.1f %% = %.1f us per window\n' % (nwin, wall / 1e3, t_none / 1e3, 100 * t_none / wall, t_none / max(nwin, 1)))
    f.write('# what the two stage-2 queues run meanwhile (sequence position:kernel; us per window)\n')
    for key, v in sorted(acc.items(), key=lambda kv: -kv[1])[:28]:
        f.write('%8.1f  %s\n' % (v / max(nwin, 1), '  +  '.join(key)))
    f.write('# per sequence position of a window on a stage-2 queue: average duration, average wait since the previous kernel of the queue ended (us)\n')
    for pos in sorted(per_pos):
        n, d, w, name = per_pos[pos]
        f.write('%02d %-44s n=%-5d dur %8.1f  wait %7.1f\n' % (pos, name, n, d / n, w / n))
print(open(sys.argv[2]).read())
