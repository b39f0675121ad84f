#!/bin/bash
# A/B of the input-patch variants: RY_PATCH bit 0 = deconvolution layers, bit 1 = k4 s2 convolution layers
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do
for v in 3 1; do
echo "RY_PATCH=$v : $(RY_PATCH=$v python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['graph_replay_ms'])")"
done; done
bash scripts/gpu_layers.sh 1 | grep encoder
