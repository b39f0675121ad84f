#!/bin/bash
# A/B of environment switches on one box, interleaved: usage gpu_r2_ab.sh "<VAR=val ...>" "<VAR=val ...>" [reps]
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/ab; mkdir -p $O; export TMPDIR=/tmp
A="$1"; B="$2"; R=${3:-3}
for i in $(seq 1 $R); do for tag in A B; do
  if [ $tag = A ]; then E="$A"; else E="$B"; fi
  env $E timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 40 > $O/${tag}_$i.json 2> $O/${tag}_$i.err
  python - <<PY
import json; d=json.load(open('$O/${tag}_$i.json')); k=d['kernels']
print('$tag [$E]', d['ms_per_step'], d['graph_replay_ms'], 'sr_last', [v['ms'] for n,v in k.items() if n.startswith('ry_sr_last')], 'pad', [v['ms'] for n,v in k.items() if n.startswith('ry_pad')], 'first', [v['ms'] for n,v in k.items() if n.startswith('ry_sr_first')])
PY
done; done
