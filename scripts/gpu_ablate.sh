#!/bin/bash
# repeated default bench lines (A/B against another build: run the same script on both)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for i in 1 2 3 4; do
echo "bench: $(python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['graph_replay_ms'], d['roofline']['kernel'], d['roofline']['achieved'])")"
done
