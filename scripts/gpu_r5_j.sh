#!/bin/bash
# Round 5, GPU call J: the whole GPU suite on the current tree (cooperative stage-1 minimum, poison test, parity corners), three times the flaky
# selection of call I to be sure the bf16 chained test is stable again, then the default bench line.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_j; mkdir -p $O
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "trained or lanes or config5 or chained or window" 2>&1 | tail -1; done
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest gpu exit $?"; tail -6 $O/pytest_gpu.txt
timeout 400 python bench.py --layers-out $O/layers.txt --details-out $O/bench_details.json > $O/bench_default.json 2> $O/bench_default.err; echo "bench exit $?"; head -c 600 $O/bench_default.json; echo
python - <<'PY'
import json
d = json.load(open('gpurun_out/r5_j/bench_default.json'))
print({k: d.get(k) for k in ('value', 'ms_per_step', 'spread', 'graph_replay_ms', 'small_window', 'mixed_stream')})
print(d['roofline']); print(d.get('roofline_stage2_forward')); print(d.get('roofline_stage1')); print(d.get('cpu_baseline'))
PY
