#!/bin/bash
# A/B of the XCD grouping (M-tile groups x filter-slice groups) against contiguous runs per XCD
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for i in 1 2 3; do
for v in 1 0; do
echo "RY_XCD_GROUPS=$v: $(RY_XCD_GROUPS=$v python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['graph_replay_ms'], d['roofline']['achieved'])")"
done; done
bash scripts/gpu_layers.sh 1 | grep -v "reduce\|sr_\|^=="
