#!/bin/bash
# sample sclk / power while the bench loop is running (is the MFMA peak clock actually sustained?)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|mclk" | head -5
python bench.py --steps 6000 --warmup 50 --no-cpu-baseline > gpurun_out/clk_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|mclk" | tr '\n' ' '; echo; sleep 0.7; done
wait $BP
cat gpurun_out/clk_bench.json | cut -c1-300
