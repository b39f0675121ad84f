#!/bin/bash
# Round 5, GPU call F: the planner's output-stationary picks (cost model) against the implicit GEMM at 300 / 100 / 600 frames, the whole GPU
# suite on the new defaults, the default bench line with the per-layer table.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_f; mkdir -p $O
for n in 300 100 600; do SWEEP_LAYERS=none timeout 300 python scripts/gpu_r5_os_sweep.py $n $O/os_defaults_n$n.txt > $O/defaults$n.log 2>&1; echo "defaults $n exit $?"; grep -v "^# with the winners\|^# winners" $O/defaults$n.log | tail -12; done
timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; echo "pytest gpu exit $?"; tail -8 $O/pytest_gpu.txt
timeout 400 python bench.py --layers-out $O/layers.txt --details-out $O/bench_details.json > $O/bench_default.json 2> $O/bench_default.err; echo "bench exit $?"; head -c 1500 $O/bench_default.json; echo; grep stage2 $O/layers.txt
