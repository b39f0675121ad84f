#!/bin/bash
# Upper bounds before building anything: the step with parts of the forward left out (RY_IGEMM_DBG bits 16 / 32: WRONG results).
# usage (GPU box): bash scripts/gpu_r3_ablate.sh <out-dir-under-gpurun_out>
out=gpurun_out/${1:-ablate}; mkdir -p $out
for dbg in 0 16 32 48 0; do
  for lanes in 2 1; do
    RY_IGEMM_DBG=$dbg timeout 300 python bench.py --lanes $lanes --steps 200 --warmup 10 --no-cpu-baseline --no-extras --no-split-bf16 > $out/b_${dbg}_${lanes}.json 2> $out/b_${dbg}_${lanes}.err
    python - $out/b_${dbg}_${lanes}.json $dbg $lanes <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('dbg %3s lanes %s  ms_per_step %.4f' % (sys.argv[2], sys.argv[3], d['ms_per_step']))
except Exception as e:
    print('dbg %3s lanes %s  FAILED %r' % (sys.argv[2], sys.argv[3], e))
PY
  done
done | tee $out/summary.txt
