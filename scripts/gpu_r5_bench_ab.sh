#!/bin/bash
# Round 5: the planner's near-tie rule (RY_KG_SLABS, default 1) measured with bench.py itself -- fresh processes, in turn, on one box.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s; mkdir -p $O
for r in 1 2 3 4; do for v in 0 1; do
  RY_KG_SLABS=$v timeout 300 python bench.py --no-extras --no-cpu-baseline --details-out $O/d.json > $O/b_${v}_$r.json 2> $O/b_${v}_$r.err
  python - $O/b_${v}_$r.json $v $r <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('RY_KG_SLABS=%s run %s  value %9.1f  ms/step %.4f  spread %.3f  brackets %s  stage2_alone %s' % (sys.argv[2], sys.argv[3], d['value'], d['ms_per_step'], d['spread'], ' '.join('%.2f' % b['wall_ms'] for b in d['brackets']), d['graph_replay_ms']['stage2_alone']))
PY
done; done
