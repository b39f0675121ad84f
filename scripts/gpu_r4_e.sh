#!/bin/bash
# Round 4, call E: the 256 x 128 tile of the bf16 / split-bf16 implicit GEMM against the planner's tiles, layer by layer and on the step.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_e; mkdir -p $O
run() { tag=$1; plan=$2; dt=$3; fr=$4
  RY_PLAN="$plan" timeout 300 python3 bench.py --dtype $dt --frames $fr --steps 100 --no-extras --no-cpu-baseline --layers-out $O/layers_$tag.txt --details-out $O/d.json > $O/$tag.json 2> $O/$tag.err
  python3 - $O/$tag.json "$tag" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print('%-28s value %9.1f  ms/step %.4f  stage2_alone %s' % (sys.argv[2], d['value'], d['ms_per_step'], d.get('graph_replay_ms', {}).get('stage2_alone')))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
BIG="1:2:0:1,2:2:0:1,3:2:0:1,4:2:0:1,11:2:0:1,12:2:0:1,13:2:0:1"
run x3_planner "" bf16x3 300
run x3_all256 "$BIG" bf16x3 300
run x3_planner_b "" bf16x3 300
run x3_dec256 "11:2:0:1,12:2:0:1,13:2:0:1" bf16x3 300
run x3_enc256 "1:2:0:1,2:2:0:1,3:2:0:1,4:2:0:1" bf16x3 300
run bf16_planner "" bf16 400
run bf16_all256 "$BIG" bf16 400
for t in x3_planner x3_all256 bf16_planner bf16_all256; do echo "== $t"; grep "stage2" $O/layers_$t.txt | grep "igemm" | awk '{printf "%-12s %-44s %-14s %9s us %9s TF\n", $2, $3, $4, $5, $7}'; done
tail -3 $O/x3_all256.err
