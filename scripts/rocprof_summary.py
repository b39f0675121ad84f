#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace) into the per-kernel stats table committed under profiles/."""
import glob
import sqlite3
import sys


def main(db_glob, out_path, title):
    db = sorted(glob.glob(db_glob, recursive=True))[0]
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    with open(out_path, 'w') as f:
        f.write('# %s\n# source: rocprofv3 --kernel-trace --stats (rocpd top_kernels view); durations in microseconds\n' % title)
        f.write('%-72s %8s %14s %12s %8s\n' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct'))
        for name, calls, total, avg, pct in rows:
            f.write('%-72s %8d %14.1f %12.3f %8.2f\n' % (name[:72], calls, total, avg, pct))
    print(open(out_path).read())


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else 'rocprofv3 kernel stats')
