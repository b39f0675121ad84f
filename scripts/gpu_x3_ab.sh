cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/x3
run() { tag=$1; shift; env "$@" RY_X3_MINM=128 python bench.py --no-cpu-baseline --dtype bf16x3 --steps 100 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'], d['graph_replay_ms']['stage2_alone'])"; }
run base RY_PLAN=
run best3 RY_PLAN=2:6:2:1,11:1:4:1,13:1:1:1
run base RY_PLAN=
run c3c5 RY_PLAN=11:1:4:1,13:1:1:1
run c5only RY_PLAN=13:1:1:1
run c5_96 RY_PLAN=13:6:1:1
run best3 RY_PLAN=2:6:2:1,11:1:4:1,13:1:1:1
