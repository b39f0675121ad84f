#!/bin/bash
# The evidence of ONE source state: everything bench.py's roofline objects are recomputed from, keyed by the hash of csrc/.
#   gpu_profile.sh [round] [tag]  ->  gpurun_out/<round>_<tag>/{kernel_stats.txt, kernel_stats_two_lanes.txt, pmc_summary.txt, layers.txt,
#                                      overlap_lanes1.txt, overlap_lanes2.txt, bench_default.json}; the three summaries also go to profiles/<round>/<tag>_*
# Order matters: the rocprofv3 summaries are written FIRST and copied into profiles/, so that the bench line at the end finds a
# kernel_stats / pmc summary of its own source hash and reports `roofline.frac` from the rocprofv3 average (not from its own HIP events).
# Counters: separate `--pmc` passes with `--kernel-trace` only, eager launches (RY_GRAPH=0) so that every dispatch is visible.
cd "$GRAFT_REPO_ROOT"; RND=${1:-r06}; TAG=${2:-z}; O=$GRAFT_REPO_ROOT/gpurun_out/${RND}_$TAG; mkdir -p $O/prof $O/pmc profiles/$RND; export TMPDIR=/tmp
HASH=$(python -c "import bench; print(bench.source_hash())")
BOX="box $(hostname) $(date -u +%F)"      # the summaries say where and when they were measured (bench.py quotes it in roofline.frac_source)
# 1. kernel durations one window at a time (--lanes 1): the durations the roofline objects are about
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --lanes 1 --no-cpu-baseline --no-extras > $O/prof_bench.json 2> $O/prof.err; echo "rocprof stats exit $?")
python scripts/rocprof_summary.py "$O/prof/**/*.db" $O/kernel_stats.txt "$RND ($TAG, source $HASH): rocprofv3 --kernel-trace --stats -- python bench.py --lanes 1 --no-cpu-baseline --no-extras (N=300, SYN-64, 1 GPU; one window at a time); $BOX" > /dev/null 2>&1 || echo "no rocpd summary"
rm -rf $O/prof; mkdir -p $O/prof
# 2. the same with the default two lanes (durations include the sharing)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > $O/prof_bench2.json 2> $O/prof2.err; echo "rocprof stats (two lanes) exit $?")
python scripts/rocprof_summary.py "$O/prof/**/*.db" $O/kernel_stats_two_lanes.txt "$RND ($TAG, source $HASH): rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extras (N=300, SYN-64, 1 GPU; the default: two windows side by side, durations include the sharing); $BOX" > /dev/null 2>&1 || echo "no rocpd summary"
rm -rf $O/prof
# 3. PMC passes
BENCH="python $GRAFT_REPO_ROOT/bench.py --lanes 1 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-extras --profile-reps 1"
run_pass() { name=$1; shift; (cd /tmp && RY_GRAPH=0 timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc/$name -o $name -- $BENCH > $O/pmc/$name.json 2> $O/pmc/$name.err; echo "pass $name exit $?"); }
run_pass p1 GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run_pass p2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
run_pass p3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS
run_pass p4 FETCH_SIZE
run_pass p5 WRITE_SIZE
python scripts/pmc_summary.py $O/pmc $O/pmc_summary.txt "$RND ($TAG), source $HASH: rocprofv3 --kernel-trace --pmc <counters>, five separate passes, RY_GRAPH=0; command: python bench.py --lanes 1 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-extras --profile-reps 1 (N=300, SYN-64, 1 GPU); $BOX"
rm -rf $O/pmc/p*/
# 4. kernel traces of the timed steps with one and two lanes: busy time per hardware queue, time kernels of two queues run together
for L in 1 2; do
  rm -rf $O/t$L; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t$L -o tr -- python $GRAFT_REPO_ROOT/bench.py --lanes $L --steps 200 --repeats 1 --no-cpu-baseline --no-extras > $O/bench_l$L.json 2> $O/err_l$L.txt; echo "trace lanes $L exit $?")
  python scripts/overlap_summary.py "$O/t$L/**/*kernel_trace.csv" $O/overlap_lanes$L.txt "$RND (source $HASH): rocprofv3 --kernel-trace -- python bench.py --lanes $L --steps 200 --repeats 1 --no-cpu-baseline --no-extras; middle half of the dispatches (timed steps)"
  rm -rf $O/t$L
done
# 5. the driver-style line, now that the summaries of this source exist under profiles/
cp $O/kernel_stats.txt profiles/$RND/${TAG}_kernel_stats.txt; cp $O/kernel_stats_two_lanes.txt profiles/$RND/${TAG}_kernel_stats_two_lanes.txt; cp $O/pmc_summary.txt profiles/$RND/${TAG}_pmc_summary.txt
timeout 400 python bench.py --layers-out $O/layers.txt > $O/bench_default.json 2> $O/bench_default.err; echo "bench default exit $?"; head -c 900 $O/bench_default.json; echo
python -c "
import json; d = json.load(open('$O/bench_default.json')); r = d['roofline']
print('roofline:', r['kernel'], r['frac'], '|', r['frac_source'], '| events', r['frac_events'], '| traffic', r['traffic'])"
head -12 $O/pmc_summary.txt | cut -c1-200
