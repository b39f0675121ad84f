#!/bin/bash
# A/B of two builds of the product library within one box (tuning aid): B = realtime_yukarin_amd/libry355.so (current
# sources), A = realtime_yukarin_amd/libry355_headref.so.ab (another build placed there by hand; untracked).
# BENCH_ARGS adds bench.py arguments (e.g. "--dtype bf16x3").
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
L=realtime_yukarin_amd/libry355.so; cp $L /tmp/B.so; cp realtime_yukarin_amd/libry355_headref.so.ab /tmp/A.so
for rep in 1 2 3; do for v in A B; do
  cp /tmp/$v.so $L
  python bench.py --no-cpu-baseline --no-split-bf16 ${BENCH_ARGS:-} --steps 60 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v rep$rep:', d['graph_replay_ms'], 'ms/step', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['achieved'])
"
done; done
cp /tmp/B.so $L
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
