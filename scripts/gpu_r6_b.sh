#!/bin/bash
# round 6, call b: where the Winograd launches spend their time (ablations, PMC counters per kernel) + the first bench line
cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/r6_b; mkdir -p $O/pmc; export TMPDIR=/tmp
timeout 600 python scripts/gpu_r6_ablate.py 300 $O/ablate_n300.txt > $O/ablate.log 2>&1; echo "ablate exit $?"; cat $O/ablate_n300.txt
HASH=$(python -c "import bench; print(bench.source_hash())")
BENCH="python $GRAFT_REPO_ROOT/bench.py --lanes 1 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-extras --profile-reps 1"
run_pass() { name=$1; shift; (cd /tmp && RY_GRAPH=0 timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc/$name -o $name -- $BENCH > $O/pmc/$name.json 2> $O/pmc/$name.err; echo "pass $name exit $?"); }
run_pass p1 GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run_pass p2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
run_pass p3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS
run_pass p4 FETCH_SIZE
run_pass p5 WRITE_SIZE
python scripts/pmc_summary.py $O/pmc $O/pmc_summary.txt "round 6 (r6_b), source $HASH: rocprofv3 --kernel-trace --pmc <counters>, five separate passes, RY_GRAPH=0; command: python bench.py --lanes 1 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-extras --profile-reps 1 (N=300, SYN-64, 1 GPU)"
rm -rf $O/pmc/p*/
cut -c1-260 $O/pmc_summary.txt | head -40
timeout 400 python bench.py --layers-out $O/layers.txt > $O/bench_default.json 2> $O/bench_default.err; echo "bench default exit $?"; head -c 1500 $O/bench_default.json; echo
