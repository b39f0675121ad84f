#!/bin/bash
# Round 4, call B: the 100-frame window (config #3 / #4 core) -- baseline, lanes, plan sweep under two lanes -- and the bottom-filter prefetch experiment.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_b; mkdir -p $O
short() { python3 - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
g = d.get('graph_replay_ms', {})
print('   value %9.1f  ms/step %.4f  spread %.4f  chain_one_window_synced %s  stage2_alone %s  s2 frac %s' % (
    d['value'], d['ms_per_step'], d['spread'], g.get('chain_one_window_synced'), g.get('stage2_alone'), d.get('roofline_stage2_forward', {}).get('frac')))
PY
}
for L in 2 1 3 4; do
  timeout 200 python3 bench.py --frames 100 --lanes $L --steps 100 --no-extras --no-cpu-baseline --details-out $O/d.json > $O/n100_l$L.json 2> $O/n100_l$L.err; echo "n100 lanes $L exit $?"; short $O/n100_l$L.json
done
for P in 0 256 1024 0 256; do
  for L in 1 2; do
    RY_VC_PREFETCH=$P timeout 200 python3 bench.py --lanes $L --steps 100 --no-extras --no-cpu-baseline --details-out $O/d.json > $O/pf${P}_l$L.json 2> $O/pf${P}_l$L.err; echo "prefetch $P lanes $L exit $?"; short $O/pf${P}_l$L.json
  done
done
SWEEP_SPLITS=1,2,3,4,6,8 timeout 900 python3 scripts/gpu_r3_lanesweep.py 100 $O/lanesweep_n100.txt 120 > $O/lanesweep.log 2>&1; echo "lanesweep exit $?"; tail -30 $O/lanesweep_n100.txt
