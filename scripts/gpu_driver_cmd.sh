#!/bin/bash
# The driver's own command in FRESH processes: the distribution behind the driver's one line (round 4: is the first K = 20 bracket deterministically slow, or was the
# 2.31 ms step of BENCH_r03.json a transient?)  Three runs of the exact command, then NRUN - 3 with --no-extras --no-cpu-baseline (the timed
# region comes first in the process either way).  Every run prints its brackets; BENCH_DEBUG_FENCE=1 adds the per-bracket stderr lines.
#   gpu_driver_cmd.sh [NRUN] [round]  ->  gpurun_out/<round>_driver_cmd/{run_XX.json, run_XX.err, summary.txt}
cd "$GRAFT_REPO_ROOT"; N=${1:-12}; RND=${2:-r06}; export O=gpurun_out/${RND}_driver_cmd; mkdir -p $O
for i in $(seq 1 $N); do
  X=""; [ $i -gt 3 ] && X="--no-extras --no-cpu-baseline"
  BENCH_DEBUG_FENCE=1 timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 $X --details-out $O/details_$i.json > $O/run_$i.json 2> $O/run_$i.err
  echo "run $i exit $?"
done
python3 - <<'PY' > $O/summary.txt
import json, glob, os, re
O = os.environ['O']
rows = []
for f in sorted(glob.glob(O + '/run_*.json'), key=lambda s: int(re.findall(r'run_(\d+)', s)[0])):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f, 'no line', e); continue
    b = d['brackets']
    rows.append((f.split('/')[-1], d['value'], d['ms_per_step'], d['spread'], [round(x['wall_ms'] / d['steps'], 4) for x in b],
                 [round(x['enq_ms'] / d['steps'], 4) for x in b], [round(x['dev_ms'] / d['steps'], 4) for x in b]))
print('# python3 bench.py --gpus 1 --steps 20 --warmup 5 in fresh processes (runs 1-3 exact, the others with --no-extras --no-cpu-baseline)')
print('# per run: value (median bracket), ms/step, spread, then per bracket: wall ms/step | enqueue ms/step | device-event ms/step')
for r in rows:
    print('%-12s value %9.1f  ms/step %.4f  spread %.4f' % r[:4]); print('    wall', r[4]); print('    enq ', r[5]); print('    dev ', r[6])
first = [r[4][0] for r in rows]; med = [r[2] for r in rows]
if rows:
    print('first bracket ms/step: min %.4f max %.4f | median-bracket ms/step: min %.4f max %.4f | line bytes: %s'
          % (min(first), max(first), min(med), max(med), [len(open(O + '/' + r[0]).read()) for r in rows]))
PY
cat $O/summary.txt
