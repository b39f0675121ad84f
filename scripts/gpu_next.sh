#!/bin/bash
# First GPU call of the next round (everything here was prepared but not measured when round 1 ran out of GPU minutes):
#   1. the GPU suite (includes the four full-size scaling-property cases of the split-bf16 path that have not run on hardware yet)
#   2. RY_AUTOTUNE against the planner (plan-build time, stage-2 replay time, result agreement) in fp32 and split-bf16 mode
#   3. the split-bf16 plan sweeps on the final kernels (DMA pieces issued in the first K step), 300 / 100 / 400 / 1000 frames
#   4. default bench line and the split-bf16 line
# ~3 GPU-minutes.  Outputs under gpurun_out/next/.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/next; mkdir -p $O; export TMPDIR=/tmp
timeout 400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest gpu exit $?"; tail -3 $O/pytest_gpu.txt
timeout 300 python scripts/gpu_autotune.py 300 > $O/autotune_n300.txt 2>&1; echo "autotune exit $?"; tail -8 $O/autotune_n300.txt
for F in 300 100 400 1000; do
  timeout 300 python scripts/gpu_x3_plansweep.py $F $O/x3_plansweep_n$F.txt 2> $O/plansweep_$F.err | grep -v "^# split"; echo "sweep $F exit $?"
done
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench exit $?"; head -c 900 $O/bench_default.json; echo
timeout 200 python bench.py --dtype bf16x3 --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3.err; head -c 400 $O/bench_x3.json; echo
