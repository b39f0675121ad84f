#!/bin/bash
# A/B of several builds of the product library on one box, interleaved (tuning aid).  Builds are realtime_yukarin_amd/libry355_<tag>.so.ab
# (untracked; made by hand with extra -D flags); "cur" = the library of the current sources.  usage: gpu_r2_libab.sh "cur isl1 isl2" [reps]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/libab; export TMPDIR=/tmp
L=realtime_yukarin_amd/libry355.so; cp $L /tmp/cur.so
for t in $1; do [ $t = cur ] || cp realtime_yukarin_amd/libry355_$t.so.ab /tmp/$t.so; done
for rep in $(seq 1 ${2:-3}); do for t in $1; do
  cp /tmp/$t.so $L
  python bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t rep$rep:', d['graph_replay_ms'], 'ms/step', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['achieved'])
"
done; done
cp /tmp/cur.so $L
