#!/bin/bash
# split-bf16 ('bf16x3') mode on the GPU: parity first, then the bench lines (x3 / fp32 side by side), a rocprofv3 kernel trace
# of the x3 run, the row-threshold and window-size sweeps, and the whole GPU suite last.  Every step writes its own file under
# gpurun_out/ at once, so a call that runs out of box time still leaves what it finished.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/x3 gpurun_out/prof_x3; export TMPDIR=/tmp
O=gpurun_out/x3
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -s -k "x3" > $O/pytest_x3.txt 2>&1; echo "pytest x3 exit $?"; tail -4 $O/pytest_x3.txt
b() { name=$1; shift; timeout 200 python bench.py --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err; echo "$name exit $? $(python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print(d['value'], 'frames/s', d['ms_per_step'], 'ms', d['graph_replay_ms'], d['roofline']['kernel'], d['roofline']['achieved'])
except Exception as e:
    print('no json', e)
PY
)"; }
b bench_x3_n300 --dtype bf16x3 --layers-out $O/layers_x3_n300.txt
timeout 300 python bench.py --layers-out $O/layers_f32_n300.txt > $O/bench_f32_default.json 2> $O/bench_f32_default.err; echo "fp32 default exit $?"; head -c 700 $O/bench_f32_default.json; echo
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_x3" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --dtype bf16x3 > "$GRAFT_REPO_ROOT/$O/prof_bench_x3.json" 2> "$GRAFT_REPO_ROOT/$O/prof_x3.err"; echo "rocprof x3 exit $?")
python scripts/rocprof_summary.py "gpurun_out/prof_x3/**/*.db" $O/kernel_stats_x3.txt "bench.py --dtype bf16x3 (300-frame window, split-bf16 stage-2)" > /dev/null 2>&1; ls -R gpurun_out/prof_x3 | head -8
RY_X3_MINM=128 b bench_x3_n300_minm128 --dtype bf16x3 --steps 30
RY_X3_MINM=2048 b bench_x3_n300_minm2048 --dtype bf16x3 --steps 30
b bench_x3_n100 --dtype bf16x3 --frames 100 --steps 30
b bench_x3_n400 --dtype bf16x3 --frames 400 --steps 30
b bench_x3_n1000 --dtype bf16x3 --frames 1000 --steps 20
b bench_x3_n300_w8 --dtype bf16x3 --windows 8 --steps 10
b bench_bf16_n300 --dtype bf16 --steps 30
timeout 600 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_all.txt 2>&1; echo "pytest gpu exit $?"; tail -3 $O/pytest_gpu_all.txt
