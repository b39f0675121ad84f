#!/bin/bash
# Round-2 evidence of one HEAD: default bench line, rocprofv3 --kernel-trace --stats of the same command, PMC passes (own runs,
# --kernel-trace only, eager launches so that every dispatch is visible) summarised with the source hash bench.py checks.
# usage: gpu_r2_profile.sh <tag>     -> gpurun_out/<tag>/{bench_default.json, kernel_stats.txt, pmc_summary.txt, layers.txt}
cd "$GRAFT_REPO_ROOT"; TAG=${1:-r02}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O/prof $O/pmc; export TMPDIR=/tmp
HASH=$(python -c "import bench; print(bench.source_hash())")
timeout 400 python bench.py --layers-out $O/layers.txt > $O/bench_default.json 2> $O/bench_default.err; echo "bench default exit $?"; head -c 600 $O/bench_default.json; echo
# kernel durations one window at a time (--lanes 1): these are the durations the roofline objects of bench.py are about (its live per-launch
# measurement is launch by launch on one stream); with the default two lanes the kernels of two windows share the chip and every duration stretches
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --lanes 1 --no-cpu-baseline --no-extras > $O/prof_bench.json 2> $O/prof.err; echo "rocprof stats exit $?")
python scripts/rocprof_summary.py "$O/prof/**/*.db" $O/kernel_stats.txt "round 2 ($TAG, source $HASH): rocprofv3 --kernel-trace --stats -- python bench.py --lanes 1 --no-cpu-baseline --no-extras (N=300, SYN-64, 1 GPU; one window at a time)" > /dev/null 2>&1 || echo "no rocpd summary"
rm -rf $O/prof; mkdir -p $O/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > $O/prof_bench2.json 2> $O/prof2.err; echo "rocprof stats (two lanes) exit $?")
python scripts/rocprof_summary.py "$O/prof/**/*.db" $O/kernel_stats_two_lanes.txt "round 2 ($TAG, source $HASH): rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extras (N=300, SYN-64, 1 GPU; the default: two windows side by side, durations include the sharing)" > /dev/null 2>&1 || echo "no rocpd summary"
BENCH="python $GRAFT_REPO_ROOT/bench.py --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --profile-reps 1"
run_pass() { name=$1; shift; (cd /tmp && RY_GRAPH=0 timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc/$name -o $name -- $BENCH > $O/pmc/$name.json 2> $O/pmc/$name.err; echo "pass $name exit $?"); }
run_pass p1 GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run_pass p2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
run_pass p3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS
run_pass p4 FETCH_SIZE
run_pass p5 WRITE_SIZE
python scripts/pmc_summary.py $O/pmc $O/pmc_summary.txt "round 2 ($TAG), source $HASH: rocprofv3 --kernel-trace --pmc <counters>, five separate passes, RY_GRAPH=0; command: python bench.py --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --profile-reps 1 (N=300, SYN-64, 1 GPU)"
head -30 $O/pmc_summary.txt | cut -c1-200
rm -rf $O/pmc/p*/ $O/prof
