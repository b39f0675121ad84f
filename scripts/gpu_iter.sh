#!/bin/bash
# iteration loop on the GPU box: parity tests, bench at N=300 and N=100 with per-layer dumps
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
show() { python -c "
import json,sys; d=json.load(open(sys.argv[1])); print({k:d[k] for k in ('value','ms_per_step','x_realtime','stage_ms','graph_replay_ms','host_call_ms_per_window')}); print(d.get('roofline')); print(d.get('roofline_stage1')); print(d['kernels'])" $1; }
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --layers-out gpurun_out/layers_n300.txt > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"
show gpurun_out/bench.json
true
true
timeout 300 python bench.py --steps 50 --warmup 5 --frames 100 --no-cpu-baseline --layers-out gpurun_out/layers_n100.txt > gpurun_out/bench_n100.json 2>> gpurun_out/bench.err
echo "--- N=100"; show gpurun_out/bench_n100.json
timeout 300 python bench.py --steps 50 --warmup 5 --frames 400 --dtype bf16 --no-cpu-baseline --layers-out gpurun_out/layers_n400_bf16.txt > gpurun_out/bench_n400_bf16.json 2>> gpurun_out/bench.err
echo "--- N=400 bf16 (config #5)"; show gpurun_out/bench_n400_bf16.json
timeout 300 python bench.py --steps 50 --warmup 5 --frames 400 --no-cpu-baseline > gpurun_out/bench_n400.json 2>> gpurun_out/bench.err
echo "--- N=400 fp32"; show gpurun_out/bench_n400.json
grep "stage-2 log-spectrum" gpurun_out/pytest_gpu.log
cat gpurun_out/layers_n300.txt
grep -v amdgpu.ids gpurun_out/bench.err | tail -3
