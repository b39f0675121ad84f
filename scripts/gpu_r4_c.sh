#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_c; mkdir -p $O
timeout 300 python3 bench.py --frames 100 --steps 50 --no-cpu-baseline --no-split-bf16 --layers-out $O/layers_n100.txt --details-out $O/details_n100.json > $O/bench_n100.json 2> $O/bench_n100.err; echo "exit $?"
cat $O/layers_n100.txt
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --details-out $O/d.json > $O/bench_drv.json 2> $O/bench_drv.err; python3 -c "
import json; d=json.loads(open('$O/bench_drv.json').read().strip().splitlines()[-1]); print(d['value'], d['spread'], [round(b['wall_ms']/20,4) for b in d['brackets']])"
