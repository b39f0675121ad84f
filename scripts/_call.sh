cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/r06_z; mkdir -p $O; export TMPDIR=/tmp
HASH=$(python -c "import bench; print(bench.source_hash())")
for L in 1 2; do
  rm -rf $O/t$L; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t$L -o tr -- python $GRAFT_REPO_ROOT/bench.py --lanes $L --steps 200 --repeats 1 --no-cpu-baseline --no-extras > $O/bench_l$L.json 2> $O/err_l$L.txt; echo "trace lanes $L exit $?")
  python scripts/overlap_summary.py "$O/t$L/**/*kernel_trace.csv" $O/overlap_lanes$L.txt "r06 (source $HASH): rocprofv3 --kernel-trace -- python bench.py --lanes $L --steps 200 --repeats 1 --no-cpu-baseline --no-extras; middle half of the dispatches (timed steps)"
  rm -rf $O/t$L
done
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee $O/pytest_gpu.txt
