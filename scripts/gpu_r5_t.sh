#!/bin/bash
# Round 5, GPU call T: kernel trace of the two-lane bench, then scripts/lane_timeline.py: what runs while no MFMA-bound launch is on the chip
cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/r5_t; mkdir -p $O; export TMPDIR=/tmp
rm -rf $O/t2; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t2 -o tr -- python $GRAFT_REPO_ROOT/bench.py --lanes 2 --steps 200 --repeats 1 --no-cpu-baseline --no-extras > $O/bench_l2.json 2> $O/err_l2.txt; echo "trace exit $?")
python scripts/lane_timeline.py "$O/t2/**/*kernel_trace.csv" $O/lane_timeline.txt "round 5: rocprofv3 --kernel-trace -- python bench.py --lanes 2 --steps 200 --repeats 1 --no-cpu-baseline --no-extras; middle half of the dispatches" | head -80
rm -rf $O/t2
