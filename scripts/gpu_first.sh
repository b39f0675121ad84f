#!/bin/bash
# first GPU round-trip: parity tests, bench, rocprof kernel trace
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
nproc > gpurun_out/host.txt; lscpu | head -20 >> gpurun_out/host.txt; rocm-smi --showproductname >> gpurun_out/host.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -30 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --steps 30 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"
cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
timeout 300 python bench.py --steps 30 --warmup 5 --frames 100 --no-cpu-baseline > gpurun_out/bench_n100.json 2>> gpurun_out/bench.err
cat gpurun_out/bench_n100.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r1" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof.err"; echo "rocprof exit $?")
find gpurun_out/prof_r1 -name "*stats*" | head; 
