#!/bin/bash
# end-of-round artifacts of build r01_n: GPU suite, default bench line (exact fp32 headline + the split-bf16 measurement beside
# it), rocprofv3 kernel traces of the fp32 and the split-bf16 run, the window-size sweep in split-bf16 mode, and the PMC passes
# (own runs, --kernel-trace only) of the split-bf16 run.  Every step writes under gpurun_out/final/ at once.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/final; mkdir -p $O/prof_f32 $O/prof_x3 $O/pmc; export TMPDIR=/tmp
timeout 400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest gpu exit $?"; tail -2 $O/pytest_gpu.txt
timeout 300 python bench.py --layers-out $O/layers_default.txt > $O/bench_default.json 2> $O/bench_default.err; echo "bench default exit $?"; head -c 1200 $O/bench_default.json; echo
timeout 200 python bench.py --dtype bf16x3 --no-cpu-baseline --layers-out $O/layers_x3.txt > $O/bench_x3.json 2> $O/bench_x3.err; echo "bench x3 exit $?"; head -c 600 $O/bench_x3.json; echo
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_f32" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-split-bf16 > "$GRAFT_REPO_ROOT/$O/prof_bench_f32.json" 2> "$GRAFT_REPO_ROOT/$O/prof_f32.err"; echo "rocprof f32 exit $?")
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_x3" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --dtype bf16x3 > "$GRAFT_REPO_ROOT/$O/prof_bench_x3.json" 2> "$GRAFT_REPO_ROOT/$O/prof_x3.err"; echo "rocprof x3 exit $?")
python scripts/rocprof_summary.py "$O/prof_f32/**/*.db" $O/kernel_stats_f32.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-split-bf16 (300-frame window, exact fp32)" > /dev/null 2>&1
python scripts/rocprof_summary.py "$O/prof_x3/**/*.db" $O/kernel_stats_x3.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --dtype bf16x3 (300-frame window, split-bf16 stage-2)" > /dev/null 2>&1
rm -rf $O/prof_f32 $O/prof_x3
for cfg in "--frames 100" "--frames 300" "--frames 400" "--frames 600" "--frames 1000" "--frames 300 --windows 4" "--frames 300 --windows 8" "--frames 1000 --windows 8"; do
  python bench.py $cfg --dtype bf16x3 --no-cpu-baseline --steps 30 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('%-30s bf16x3 value %9.0f  x_rt %6.0f  ms/step %7.3f  s1 %.3f s2 %.3f  | %s %.1f TF(alg) frac %.3f' % ('$cfg', d['value'], d['x_realtime'], d['ms_per_step'], d['graph_replay_ms']['stage1_alone'], d['graph_replay_ms']['stage2_alone'], r['kernel'], r['achieved'], r['frac']))
" | tee -a $O/sweep_x3.txt
done
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-reps 1 --dtype bf16x3"
run_pass() { name=$1; shift; (cd /tmp && RY_GRAPH=0 timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/$O/pmc/$name" -o $name -- $BENCH > "$GRAFT_REPO_ROOT/$O/pmc/$name.json" 2> "$GRAFT_REPO_ROOT/$O/pmc/$name.err"; echo "pass $name exit $?"); }
run_pass p1 GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run_pass p4 FETCH_SIZE
run_pass p5 WRITE_SIZE
run_pass p2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
run_pass p3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS
python scripts/pmc_summary.py $O/pmc $O/pmc_summary_x3.txt "PMC passes of bench.py --dtype bf16x3 (eager launches, RY_GRAPH=0)" > /dev/null 2>&1; ls $O
# planner-constant A/B inside the split-bf16 mode (same box, interleaved with the default)
ab() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --dtype bf16x3 --steps 100 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ab $tag', d['value'], d['ms_per_step'], d['graph_replay_ms']['stage2_alone'])" | tee -a $O/ab_x3.txt; }
ab default RY_X3_MINM=128
ab kg2_1.3 RY_PLAN_X3_KG2=1.3
ab peak1400 RY_PLAN_X3_PEAK=1400
ab default RY_X3_MINM=128
ab peak1000 RY_PLAN_X3_PEAK=1000
ab forced RY_PLAN=10:3:16:1,12:1:5:1
