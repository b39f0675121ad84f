#!/bin/bash
# The chained step at every BASELINE window size (configs #3/#4, #5, #1, #2), in bf16 / split-bf16 mode and with 8 lanes: the table of DESIGN.md section 7.
#   gpu_sweep.sh [round] [tag]  ->  gpurun_out/<round>_<tag>/sweep.txt
cd "$GRAFT_REPO_ROOT"; RND=${1:-r06}; TAG=${2:-w}; O=$GRAFT_REPO_ROOT/gpurun_out/${RND}_$TAG; mkdir -p $O
HASH=$(python -c "import bench; print(bench.source_hash())")
echo "# $RND (source $HASH): python bench.py --no-cpu-baseline --no-split-bf16 <args>; one MI355X, SYN-64" > $O/sweep.txt
echo "# args | frames/s | ms per window | x real-time | effective x real-time | mixed stream frames/s | stage-2 forward alone ms | whole stage-2 forward / peak (executed FLOPs) | lone host call ms | batch8 ms per window" >> $O/sweep.txt
run() {
  timeout 300 python bench.py --no-cpu-baseline --no-split-bf16 "$@" > $O/line.json 2> $O/line.err || { echo "$* FAILED" >> $O/sweep.txt; tail -3 $O/line.err >> $O/sweep.txt; return; }
  python - "$O/line.json" "$*" >> $O/sweep.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
g = lambda *k: (lambda v: v if v is not None else float('nan'))(__import__('functools').reduce(lambda a, b: (a or {}).get(b) if isinstance(a, dict) else None, k, d))
print('%-34s | %9.0f | %.4f | %6.0f | %6.0f | %9.0f | %.4f | %.3f | %.4f | %.4f' % (
    sys.argv[2] or '(default)', d['value'], d['ms_per_step'] / d['config']['windows_per_gpu'], d['x_realtime'], d['effective_x_realtime'],
    g('mixed_stream', 'value'), g('graph_replay_ms', 'stage2_alone'), g('roofline_stage2_forward', 'frac'), g('host_path', 'call_ms_per_window'),
    g('chained_batch8', 'ms_per_window')))
PY
}
run --frames 100
run --frames 200
run
run --frames 400
run --frames 600 --extra-frames 200
run --frames 1000
run --lanes 1
run --lanes 8
run --windows 8
run --dtype bf16 --frames 400
run --dtype bf16x3
cat $O/sweep.txt
