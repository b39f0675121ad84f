#!/bin/bash
# Round 2: stage-1 output-stationary kernels (ry_c1d_os) -- parity on the GPU, then A/B against the round-1 weight-streaming path
# and a sweep of the slice heuristic.  Output: gpurun_out/r2b/
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r2b; mkdir -p $O; export TMPDIR=/tmp
timeout 500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest gpu exit $?"; tail -3 $O/pytest_gpu.txt
S='--no-cpu-baseline --no-split-bf16 --steps 30'
for U in 256 128 512 1024; do
  RY_S1_UNITS=$U timeout 200 python bench.py $S --layers-out $O/layers_u$U.txt > $O/bench_u$U.json 2> $O/bench_u$U.err
  python - <<PY
import json; d=json.load(open('$O/bench_u$U.json')); print('units $U', d['ms_per_step'], d['graph_replay_ms'], d['roofline_stage1']['kernel_ms_per_forward'])
PY
done
RY_S1_OS=0 timeout 200 python bench.py $S > $O/bench_ws.json 2> $O/bench_ws.err
python - <<PY
import json; d=json.load(open('$O/bench_ws.json')); print('ws', d['ms_per_step'], d['graph_replay_ms'])
PY
grep stage1 $O/layers_u256.txt
