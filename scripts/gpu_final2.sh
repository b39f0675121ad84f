#!/bin/bash
# last call of the round: A/B of the bf16 DMA issue placement (A = all four K steps, B = first K step = default) on one box,
# the GPU suite with the default build, then the bench lines that the documentation quotes
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/final3; mkdir -p $O; export TMPDIR=/tmp
L=realtime_yukarin_amd/libry355.so; cp $L /tmp/B.so; cp realtime_yukarin_amd/libry355_headref.so.ab /tmp/A.so
for rep in 1 2 3; do for v in A B; do
  cp /tmp/$v.so $L
  python bench.py --no-cpu-baseline --no-split-bf16 --dtype bf16x3 --steps 60 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v rep$rep:', d['graph_replay_ms']['stage2_alone'], 'ms/step', d['ms_per_step'], d['roofline']['achieved'])
" | tee -a $O/ab_issue.txt
done; done
cp /tmp/B.so $L
timeout 300 python -m pytest tests -x -q -m gpu 2>&1 | tail -2 | tee $O/pytest_gpu.txt
timeout 200 python bench.py --layers-out $O/layers_default.txt > $O/bench_default.json 2> $O/err1.txt; head -c 900 $O/bench_default.json; echo
timeout 100 python bench.py --dtype bf16x3 --no-cpu-baseline --layers-out $O/layers_x3.txt > $O/bench_x3.json 2> $O/err2.txt
for cfg in "--frames 100 --dtype bf16x3" "--frames 400 --dtype bf16x3" "--frames 1000 --dtype bf16x3" "--frames 300 --windows 8 --dtype bf16x3" "--frames 400 --dtype bf16" "--frames 1000 --dtype bf16"; do
  python bench.py $cfg --no-cpu-baseline --steps 30 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-40s value %9.0f ms/step %7.3f s2 %.3f | %s %.1f' % ('$cfg', d['value'], d['ms_per_step'], d['graph_replay_ms']['stage2_alone'], r['kernel'], r['achieved']))
" | tee -a $O/sweep.txt
done
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_x3" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --dtype bf16x3 > /dev/null 2> "$GRAFT_REPO_ROOT/$O/prof_x3.err"; echo "rocprof x3 exit $?")
python scripts/rocprof_summary.py "$O/prof_x3/**/*.db" $O/kernel_stats_x3.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --dtype bf16x3 (300-frame window, split-bf16 stage-2)" > /dev/null 2>&1; rm -rf $O/prof_x3
