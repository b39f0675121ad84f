#!/bin/bash
# iteration loop on the GPU box: parity tests, bench at N=300 and N=100 with per-layer dumps
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --layers-out gpurun_out/layers_n300.txt > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"
python -c "
import json; d=json.load(open('gpurun_out/bench.json')); print({k:d[k] for k in ('value','ms_per_step','x_realtime','roofline','roofline_stage1','stage_ms')}); print(d['kernels'])"
timeout 300 python bench.py --steps 50 --warmup 5 --frames 100 --no-cpu-baseline --layers-out gpurun_out/layers_n100.txt > gpurun_out/bench_n100.json 2>> gpurun_out/bench.err
python -c "
import json; d=json.load(open('gpurun_out/bench_n100.json')); print({k:d[k] for k in ('value','ms_per_step','x_realtime','stage_ms')})"
cat gpurun_out/layers_n300.txt
tail -3 gpurun_out/bench.err
