#!/bin/bash
# hardware counters (own passes, --kernel-trace only) + MFMA peak probe
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
./tools/mfma_peak > gpurun_out/mfma_peak.txt 2>&1; cat gpurun_out/mfma_peak.txt
rocprofv3 -L > gpurun_out/pmc/counters_list.txt 2>&1
grep -c . gpurun_out/pmc/counters_list.txt
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-reps 1"
run_pass() { name=$1; shift; (cd /tmp && RY_GRAPH=0 timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc/$name" -o $name -- $BENCH > "$GRAFT_REPO_ROOT/gpurun_out/pmc/$name.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/pmc/$name.err"; echo "pass $name exit $?"); }
run_pass p1 GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run_pass p2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
run_pass p3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS
run_pass p4 FETCH_SIZE
run_pass p5 WRITE_SIZE
find gpurun_out/pmc -name "*.csv" | head -20
for f in gpurun_out/pmc/*.err; do echo "== $f"; grep -v amdgpu.ids $f | tail -3; done
