#!/bin/bash
# Round 4 experiment: fp32 patch loops with the wave-private filter columns loaded global -> registers (RY_F32_BREG=1 build) against the product build.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_breg; mkdir -p $O; P=realtime_yukarin_amd
cp $P/libry355.so $O/lib_base.so; cp $P/libry355_breg.so $O/lib_breg.so
run() { tag=$1; lib=$2; shift 2
  cp $O/lib_$lib.so $P/libry355.so
  timeout 300 python3 bench.py --steps 100 --no-extras "$@" --layers-out $O/layers_$tag.txt --details-out $O/d.json > $O/$tag.json 2> $O/$tag.err
  python3 - $O/$tag.json "$tag" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    c = d.get('cpu_baseline', {}).get('gpu_result_vs_this_baseline')
    print('%-22s value %9.1f  ms/step %.4f  spread %.3f  stage2_alone %s  chain %s  parity %s' % (sys.argv[2], d['value'], d['ms_per_step'], d['spread'], d['graph_replay_ms']['stage2_alone'], d['graph_replay_ms']['chain_one_window_synced'], c))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run base_a base --no-cpu-baseline
run breg_a breg --cpu-seconds 2
run base_b base --no-cpu-baseline
run breg_b breg --no-cpu-baseline
run base_l1 base --no-cpu-baseline --lanes 1
run breg_l1 breg --no-cpu-baseline --lanes 1
run base_n100 base --no-cpu-baseline --frames 100
run breg_n100 breg --no-cpu-baseline --frames 100
cp $O/lib_base.so $P/libry355.so
for t in base_a breg_a; do echo "== $t"; grep "stage2" $O/layers_$t.txt | grep "igemm" | awk '{printf "%-12s %-44s %-14s %9s us %9s TF\n", $2, $3, $4, $5, $7}'; done
tail -2 $O/breg_a.err
