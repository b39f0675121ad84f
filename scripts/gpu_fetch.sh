#!/bin/bash
# FETCH_SIZE (L2 miss traffic) per launch of the implicit-GEMM kernels with and without the XCD grouping (own --pmc pass each)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/fetch; export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-reps 1"
for v in 1 0; do
(cd /tmp && RY_XCD_GROUPS=$v RY_GRAPH=0 timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/fetch/g$v" -o g$v -- $BENCH > /dev/null 2>&1)
python - <<PY
import csv, glob
from collections import defaultdict
d = defaultdict(list)
for path in glob.glob('gpurun_out/fetch/g$v/*counter_collection.csv'):
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == 'FETCH_SIZE' and 'ry_igemm_ldsdma' in r['Kernel_Name']:
            d[r['Kernel_Name'].split('(')[0].replace('void ', '').replace(' ', '')].append(float(r['Counter_Value']))
print('RY_XCD_GROUPS=$v')
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print('  %-48s launches %4d  FETCH_SIZE avg %9.0f KB  total %8.1f MB' % (k, len(v), sum(v) / len(v), sum(v) / 1024))
print('  all igemm launches: %.1f MB' % (sum(sum(v) for v in d.values()) / 1024))
PY
done
