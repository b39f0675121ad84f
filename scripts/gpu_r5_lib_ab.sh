#!/bin/bash
# Round 5: the product library against an experiment build (libry355<suffix>.so, build.build_product(defs=..., suffix=...)) with bench.py itself,
# fresh processes in turn on one box.   gpu_r5_lib_ab.sh <suffix> [pairs]
cd "$GRAFT_REPO_ROOT"; SUF=$1; N=${2:-3}; O=gpurun_out/r5_lib_ab$SUF; mkdir -p $O; P=realtime_yukarin_amd
cp $P/libry355.so $O/lib_base.so; cp $P/libry355$SUF.so $O/lib_exp.so
for r in $(seq 1 $N); do for v in base exp; do
  cp $O/lib_$v.so $P/libry355.so
  timeout 300 python bench.py --no-extras --no-cpu-baseline --layers-out $O/layers_$v.txt --details-out $O/d.json > $O/b_${v}_$r.json 2> $O/b_${v}_$r.err
  python - $O/b_${v}_$r.json $v $r <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('%-5s run %s  value %9.1f  ms/step %.4f  spread %.3f  brackets %s  stage2_alone %s chain %s' % (sys.argv[2], sys.argv[3], d['value'], d['ms_per_step'], d['spread'], ' '.join('%.2f' % b['wall_ms'] for b in d['brackets']), d['graph_replay_ms']['stage2_alone'], d['graph_replay_ms']['chain_one_window_synced']))
PY
done; done
cp $O/lib_base.so $P/libry355.so
grep "sr_last\|sr_first" $O/layers_base.txt $O/layers_exp.txt | cut -c1-160
