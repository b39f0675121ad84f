#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
show() { python -c "
import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], {k:d[k] for k in ('value','ms_per_step','x_realtime','graph_replay_ms')}, d['roofline']['kernel'], d['roofline']['achieved'])" $1 "$2"; }
for w in 1 2 4 8; do
timeout 300 python bench.py --steps 30 --warmup 3 --windows $w --no-cpu-baseline > gpurun_out/bench_w$w.json 2>> gpurun_out/bench.err; show gpurun_out/bench_w$w.json "windows=$w N=300 fp32:"
done
timeout 300 python bench.py --steps 30 --warmup 3 --windows 4 --frames 100 --no-cpu-baseline > gpurun_out/bench_w4_n100.json 2>> gpurun_out/bench.err; show gpurun_out/bench_w4_n100.json "windows=4 N=100 fp32:"
timeout 300 python bench.py --steps 30 --warmup 3 --windows 4 --frames 400 --dtype bf16 --no-cpu-baseline > gpurun_out/bench_w4_n400_bf16.json 2>> gpurun_out/bench.err; show gpurun_out/bench_w4_n400_bf16.json "windows=4 N=400 bf16:"
