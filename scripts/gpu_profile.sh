#!/bin/bash
# end-of-round artifacts: rocprofv3 kernel-trace stats of the default bench command + PMC passes (own runs, --kernel-trace only)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/prof gpurun_out/pmc; export TMPDIR=/tmp
timeout 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench default exit $?"; cat gpurun_out/bench_default.json | head -c 1500; echo
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof.err"; echo "rocprof stats exit $?")
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-reps 1"
run_pass() { name=$1; shift; (cd /tmp && RY_GRAPH=0 timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc/$name" -o $name -- $BENCH > "$GRAFT_REPO_ROOT/gpurun_out/pmc/$name.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/pmc/$name.err"; echo "pass $name exit $?"); }
rm -rf gpurun_out/pmc/p*
run_pass p1 GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run_pass p2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
run_pass p3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS
run_pass p4 FETCH_SIZE
run_pass p5 WRITE_SIZE
ls gpurun_out/prof | head
