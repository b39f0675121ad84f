"""A readable excerpt of a rocprofv3 --kernel-trace CSV: the dispatches of a few milliseconds in the middle of the run, one line each
(start and end in microseconds from the first one shown, duration, hardware queue, kernel), stage-2 launches of at least 40 us marked.
usage: trace_excerpt.py <kernel_trace.csv glob> <out.txt> [milliseconds]"""
import csv, glob, sys

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
span = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 5e6
mid = rows[len(rows) // 2][0]
sel = [r for r in rows if mid <= r[0] < mid + span]
t0 = sel[0][0]
with open(sys.argv[2], 'w') as f:
    f.write('# start_us end_us dur_us queue kernel   (* = MFMA-bound launch)\n')
    for s, e, q, n in sel:
        n = n.replace('void ', '').replace('(RyIgemmParams)', '').replace(' ', '')
        n = n[:n.find('(')] if '(' in n else n
        f.write('%9.1f %9.1f %7.1f  q%-3s %s%s\n' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, '* ' if ('ry_igemm' in n and e - s >= 40000) else '  ', n))
print(open(sys.argv[2]).read()[:200])
